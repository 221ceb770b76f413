// lifecycle.cpp -- MedicalSeg::initialize_engine / process_single_image / cleanup_resources on top of the C-ABI
// (include/mi_unet.h).  Reference: src/initialize.cpp:26-91, src/process.cpp:123-262, src/cleanup.cpp:10-64.
// Same log file name, banner lines, message prefixes and bool/void error conventions.  What replaces what:
//   g_runtime / g_engine (one deserialised TensorRT engine, src/initialize.cpp:20-21)
//        -> one mi_unet group: an engine handle per visible device, weights packed once and sent device-to-device
//   thread_local TensorRTContext (exec context + buffers + stream + graph per calling thread, src/process.cpp:15, :45-120)
//        -> a thread_local clone of the first device's handle (mi_unet_clone: shares the weight blob, owns buffers, stream
//           and graphs), created lazily on a thread's first single-image call, so concurrent callers do not serialise
//   the sequential file loop of directory mode (src/main.cpp:148-164)
//        -> process_image_batch: chunks of max_batch x devices images, sharded over the group
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <future>
#include <memory>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>

#include "../../include/medseg/cleanup.h"
#include "../../include/medseg/initialize.h"
#include "../../include/medseg/mask2polygon.h"
#include "../../include/medseg/postprocess.h"
#include "../../include/medseg/preprocess.h"
#include "../../include/medseg/process.h"
#include "png_io.h"

namespace fs = std::filesystem;
using medseg::Image8;

namespace MedicalSeg {

namespace {
std::mutex g_state_mutex;          // guards the four below
mi_unet_group_t *g_group = nullptr;
mi_unet_group_t *g_lane2 = nullptr;  // a clone of g_group, created by the first directory-mode call with more than one chunk
mi_unet_config g_cfg{};            // per-rank configuration of the group (tile size, topology, max_batch, algorithm)
int g_thread_batch = 1;            // micro-batch capacity of a per-thread context
unsigned long g_generation = 0;    // bumped by every (re)initialisation and cleanup: older thread contexts are stale
std::ofstream g_log_file;
std::string g_log_path;
std::mutex g_log_mutex;            // the reference's global log stream is written from any thread unguarded
std::mutex g_batch_mutex;          // the batch routes toggle group-wide state (postprocess flag): one batch call at a time

// The calling thread's context.  Destroyed by cleanup_resources() on that thread (as the reference does, src/cleanup.cpp:16)
// or when the thread ends; a context of an older engine generation is replaced on its next use.
struct ThreadContext {
    mi_unet_t *h = nullptr;
    unsigned long generation = 0;
    void release() { if (h) { mi_unet_destroy(h); h = nullptr; } }
    ~ThreadContext() { release(); }
};
thread_local ThreadContext t_context;

int env_int(const char *name, int fallback)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return fallback;
    char *end = nullptr;
    const long x = std::strtol(v, &end, 10);
    return (end && *end == '\0') ? (int)x : fallback;
}

int env_algo()
{
    const char *v = std::getenv("MEDSEG_CONV_ALGO");
    if (!v) return MI_UNET_CONV_AUTO;
    const std::string s(v);
    return s == "direct" ? MI_UNET_CONV_DIRECT : s == "winograd" ? MI_UNET_CONV_WINOGRAD : s == "bf16" ? MI_UNET_CONV_BF16
         : s == "fp16" ? MI_UNET_CONV_FP16 : MI_UNET_CONV_AUTO;
}

// topology of a MIUNETW1 file (miunet/spec.py): magic[8], u32 version, in_ch, base, levels, classes
bool read_weight_header(const std::string &path, mi_unet_config &cfg)
{
    std::ifstream f(path, std::ios::binary);
    unsigned char hdr[28];
    if (!f.read(reinterpret_cast<char *>(hdr), sizeof hdr) || std::memcmp(hdr, "MIUNETW1", 8) != 0) return false;
    uint32_t v[5];
    std::memcpy(v, hdr + 8, sizeof v);
    if (v[0] != 1) return false;
    cfg.in_ch = (int)v[1]; cfg.base = (int)v[2]; cfg.levels = (int)v[3]; cfg.classes = (int)v[4];
    return true;
}
}  // namespace

// Engine configuration: the topology comes from the weight file's header, everything the reference hard-codes or leaves to
// TensorRT comes from the environment --
//   MEDSEG_TILE_SIZE (or MEDSEG_TILE_W / MEDSEG_TILE_H)  network tile, default 512 (src/process.cpp:70, src/preprocess.cpp:81)
//   MEDSEG_MAX_BATCH     images per device micro-batch in directory mode, default 16
//   MEDSEG_THREAD_BATCH  capacity of a per-thread context (single-image calls), default 1
//   MEDSEG_CONV_ALGO     auto | direct | winograd | bf16 | fp16 (the arithmetic of BASELINE configs 3 and 5)
//   MEDSEG_DEVICES       number of devices in the group: default 1 (the reference's implicit device 0); N > 1 or 0 (= every
//                        visible device) is opt-in -- the multi-device transports have not met a second GPU yet
bool initialize_engine(const std::string &trt_cache_path, const std::string &log_dir)
{
    std::lock_guard<std::mutex> state_lock(g_state_mutex);
    try {
        fs::create_directories(log_dir);
        g_log_path = log_dir + "/segmentation_log.txt";
        if (g_log_file.is_open()) g_log_file.close();
        g_log_file.open(g_log_path, std::ios::out | std::ios::trunc);
        if (!g_log_file.is_open()) {
            std::cerr << "Failed to create log file: " << g_log_path << std::endl;
            return false;
        }
        g_log_file << "=== Initializing Medical Image Segmentation Engine ===" << std::endl;
        g_log_file << "MI355X UNet weight file: " << trt_cache_path << std::endl;
        if (!fs::exists(trt_cache_path)) {
            g_log_file << "Error: engine weight file not found - " << trt_cache_path << std::endl;
            return false;
        }
        if (g_lane2) { mi_unet_group_destroy(g_lane2); g_lane2 = nullptr; }
        if (g_group) { mi_unet_group_destroy(g_group); g_group = nullptr; }
        ++g_generation;
        mi_unet_default_config(&g_cfg);            // 512x512x1, 3 classes (src/process.cpp:70, :162)
        if (!read_weight_header(trt_cache_path, g_cfg)) {
            g_log_file << "Error: not a MIUNETW1 weight file - " << trt_cache_path << std::endl;
            std::cerr << "Initialization error: not a MIUNETW1 weight file" << std::endl;
            return false;
        }
        const int tile = env_int("MEDSEG_TILE_SIZE", 512);
        g_cfg.width = env_int("MEDSEG_TILE_W", tile);
        g_cfg.height = env_int("MEDSEG_TILE_H", tile);
        g_cfg.max_batch = std::max(1, env_int("MEDSEG_MAX_BATCH", 16));
        g_cfg.conv_algo = env_algo();
        g_thread_batch = std::max(1, env_int("MEDSEG_THREAD_BATCH", 1));
        const int n_devices = env_int("MEDSEG_DEVICES", 1);
        g_log_file << "Device group requested: " << (n_devices <= 0 ? std::string("every visible device") : std::to_string(n_devices))
                   << " (MEDSEG_DEVICES)" << std::endl;
        bool up = mi_unet_group_create(&g_cfg, nullptr, n_devices, &g_group) == MI_UNET_OK &&
                  mi_unet_group_load_weights(g_group, trt_cache_path.c_str()) == MI_UNET_OK;
        if (!up && n_devices != 1 && mi_unet_device_count() > 1) {
            // a multi-device group that does not come up must not take single-device operation with it
            g_log_file << "Warning: multi-device group failed (" << mi_unet_last_error() << "); continuing on device "
                       << g_cfg.device << " alone" << std::endl;
            if (g_group) { mi_unet_group_destroy(g_group); g_group = nullptr; }
            up = mi_unet_group_create(&g_cfg, nullptr, 1, &g_group) == MI_UNET_OK &&
                 mi_unet_group_load_weights(g_group, trt_cache_path.c_str()) == MI_UNET_OK;
        }
        if (!up) {
            g_log_file << "Error: Failed to initialize MI355X UNet engine: " << mi_unet_last_error() << std::endl;
            std::cerr << "Initialization error: " << mi_unet_last_error() << std::endl;
            if (g_group) { mi_unet_group_destroy(g_group); g_group = nullptr; }
            return false;
        }
        g_log_file << "MI355X UNet engine initialized successfully" << std::endl;
        g_log_file << "  Topology: in_ch=" << g_cfg.in_ch << " base=" << g_cfg.base << " levels=" << g_cfg.levels
                   << " classes=" << g_cfg.classes << ", tile " << g_cfg.width << "x" << g_cfg.height << std::endl;
        g_log_file << "  Devices: " << mi_unet_group_size(g_group) << " (weights to ranks > 0 by "
                   << mi_unet_group_weight_transport(g_group) << "), micro-batch " << g_cfg.max_batch << " per device" << std::endl;
        g_log_file << "  " << mi_unet_numeric_guard(mi_unet_group_handle(g_group, 0), nullptr, nullptr) << std::endl;
        g_log_file << "  Input size: " << (size_t)g_cfg.height * g_cfg.width * g_cfg.in_ch << " bytes (u8)" << std::endl;
        g_log_file << "  Output size: " << (size_t)g_cfg.height * g_cfg.width << " bytes (classes=" << g_cfg.classes << ")" << std::endl;
        return true;
    } catch (const std::exception &e) {
        std::cerr << "Initialization error: " << e.what() << std::endl;
        if (g_log_file.is_open()) g_log_file << "Initialization error: " << e.what() << std::endl;
        return false;
    }
}

mi_unet_t *get_engine()
{
    std::lock_guard<std::mutex> lk(g_state_mutex);
    return g_group ? mi_unet_group_handle(g_group, 0) : nullptr;
}
mi_unet_group_t *get_engine_group() { std::lock_guard<std::mutex> lk(g_state_mutex); return g_group; }
std::ofstream &get_log_file() { return g_log_file; }
std::string get_log_path() { return g_log_path; }

// The reference's get_thread_local_context() + initialize_context() (src/process.cpp:17-19, :45-120): the calling thread's
// own context, created on first use.
mi_unet_t *get_thread_local_context()
{
    std::lock_guard<std::mutex> lk(g_state_mutex);
    if (!g_group) throw std::runtime_error("Engine not initialized");
    if (t_context.h && t_context.generation == g_generation) return t_context.h;
    t_context.release();
    if (mi_unet_clone(mi_unet_group_handle(g_group, 0), g_thread_batch, &t_context.h) != MI_UNET_OK)
        throw std::runtime_error(std::string("context creation failed: ") + mi_unet_last_error());
    t_context.generation = g_generation;
    {
        std::lock_guard<std::mutex> ll(g_log_mutex);
        if (g_log_file.is_open())
            g_log_file << "Execution context created for a new thread (micro-batch " << g_thread_batch << ")" << std::endl;
    }
    return t_context.h;
}

namespace {
// single-plane tiles / RAW images feed every input channel of a multi-channel engine: the grey -> B,G,R replication of
// cv::imread(IMREAD_COLOR) (src/mask2polygon.cpp:117); see mi_unet_infer_raw16 in include/mi_unet.h
std::vector<uint8_t> interleave_gray(const std::vector<const Image8 *> &imgs, int in_ch)
{
    const size_t hw = (size_t)g_cfg.height * g_cfg.width;
    std::vector<uint8_t> in(hw * in_ch * imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) {
        const Image8 &g = *imgs[i];
        if (g.rows != g_cfg.height || g.cols != g_cfg.width || g.channels != 1)
            throw std::runtime_error("Input size must be " + std::to_string(g_cfg.width) + "x" + std::to_string(g_cfg.height) +
                                     " for fixed context");                                   // src/process.cpp:127
        uint8_t *dst = in.data() + i * hw * in_ch;
        if (in_ch == 1) std::copy(g.data.begin(), g.data.end(), dst);
        else
            for (size_t p = 0; p < hw; ++p)
                for (int c = 0; c < in_ch; ++c) dst[p * in_ch + c] = g.data[p];
    }
    return in;
}
}  // namespace

std::vector<Image8> execute_inference_batch(const std::vector<Image8> &gray_imgs)
{
    try {
        mi_unet_group_t *group = get_engine_group();
        if (!group) throw std::runtime_error("Engine not initialized");
        const size_t hw = (size_t)g_cfg.height * g_cfg.width;
        std::vector<const Image8 *> ptrs;
        for (const Image8 &g : gray_imgs) ptrs.push_back(&g);
        const std::vector<uint8_t> in = interleave_gray(ptrs, g_cfg.in_ch);
        std::vector<uint8_t> out(hw * gray_imgs.size());
        {
            std::lock_guard<std::mutex> lk(g_batch_mutex);
            if (mi_unet_group_infer_u8(group, in.data(), (int)gray_imgs.size(), out.data(), nullptr) != MI_UNET_OK)
                throw std::runtime_error(mi_unet_last_error());
        }
        std::vector<Image8> masks;
        masks.reserve(gray_imgs.size());
        for (size_t i = 0; i < gray_imgs.size(); ++i) {
            Image8 m(g_cfg.height, g_cfg.width, 1);
            std::copy(out.begin() + i * hw, out.begin() + (i + 1) * hw, m.data.begin());
            masks.push_back(std::move(m));
        }
        return masks;
    } catch (const std::exception &e) {
        throw std::runtime_error("Inference failed: " + std::string(e.what()));               // src/process.cpp:173
    }
}

// One tile on the calling thread's own context (src/process.cpp:123-175).
Image8 execute_inference(const Image8 &gray_img)
{
    try {
        mi_unet_t *ctx = get_thread_local_context();
        const std::vector<uint8_t> in = interleave_gray({ &gray_img }, g_cfg.in_ch);
        Image8 mask(g_cfg.height, g_cfg.width, 1);
        if (mi_unet_infer_u8(ctx, in.data(), 1, mask.data.data(), nullptr) != MI_UNET_OK) throw std::runtime_error(mi_unet_last_error());
        return mask;
    } catch (const std::exception &e) {
        throw std::runtime_error("Inference failed: " + std::string(e.what()));               // src/process.cpp:173
    }
}

Image8 mask_to_image(const Image8 &mask)
{
    uint8_t lut[256] = { 0 };
    lut[1] = 128;
    lut[2] = 255;
    Image8 vis(mask.rows, mask.cols, 1);
    for (size_t i = 0; i < mask.data.size(); ++i) vis.data[i] = lut[mask.data[i]];
    return vis;
}

namespace {

// One image after the device work is done: write the reference's artefacts and run the CPU tail of the pipeline.
void finish_image(const std::string &raw_path, int width, int height, const std::string &output_dir, const Image8 &tile,
                  Image8 pred_mask, bool already_postprocessed)
{
    const std::string base_name = fs::path(raw_path).stem().string();
    const std::string preprocessed_png_path = output_dir + "/" + base_name + "_normalized.png";
    const std::string size_json_path = output_dir + "/" + base_name + "_original_sizes.json";
    const std::string pred_mask_path = output_dir + "/" + base_name + "_mask.png";
    if (!Preprocess::write_preprocess_outputs(tile, raw_path, preprocessed_png_path, size_json_path, width, height))
        throw std::runtime_error("Preprocessing failed");
    if (!already_postprocessed) pred_mask = postprocess_mask(pred_mask);
    if (!medseg::write_png(pred_mask_path, mask_to_image(pred_mask), /*level0=*/true))
        throw std::runtime_error("Failed to save mask");
    Mask2Polygon::process_single_mask(pred_mask_path, output_dir, size_json_path, preprocessed_png_path, base_name);
}

bool host_preprocess_requested()
{
    const char *e = std::getenv("MEDSEG_HOST_PREPROCESS");
    return e && e[0] == '1';
}

// extract_contours runs on the device behind postprocess_mask unless MEDSEG_HOST_CONTOURS=1
bool device_contours_requested()
{
    const char *e = std::getenv("MEDSEG_HOST_CONTOURS");
    return !(e && e[0] == '1');
}

// postprocess_mask runs on the device right behind the argmax unless MEDSEG_HOST_POSTPROCESS=1
bool device_postprocess_requested()
{
    const char *e = std::getenv("MEDSEG_HOST_POSTPROCESS");
    return !(e && e[0] == '1');
}

}  // namespace

// ---- directory mode as a three-stage pipeline over chunks of max_batch images (all-device route):
//        read the files of chunk k+1  ||  device: chunk k (mi_unet_segment_raw16)  ||  PNG / JSON artefacts of chunk k-1
// Each stage is internally parallel over its images (a few host threads; the device call is one micro-batch); console and
// log text is collected per image and emitted in file order, chunk after chunk, by the calling thread.
namespace {

// Page-locked buffers for the RAW files of directory mode (mi_unet_host_alloc): the reader threads copy page cache -> pinned,
// the engine's DMA reads them directly, and the device thread no longer pays a staging memcpy per image.  Pinning memory is
// slow (a millisecond per 6 MB), so buffers are recycled across chunks and calls and released by cleanup_resources().
class PinnedPool {
public:
    struct Buf { uint16_t *p = nullptr; size_t cap = 0; };
    Buf acquire(size_t samples)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].cap >= samples) { Buf b = free_[i]; free_.erase(free_.begin() + i); return b; }
        }
        Buf b;
        void *p = nullptr;
        if (mi_unet_host_alloc(samples * sizeof(uint16_t), &p) != MI_UNET_OK) throw std::runtime_error(std::string("pinned allocation failed: ") + mi_unet_last_error());
        b.p = static_cast<uint16_t *>(p); b.cap = samples;
        return b;
    }
    void release(Buf b)
    {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(m_);
        free_.push_back(b);
    }
    void clear()
    {
        std::lock_guard<std::mutex> lk(m_);
        for (Buf &b : free_) mi_unet_host_free(b.p);
        free_.clear();
    }

private:
    std::mutex m_;
    std::vector<Buf> free_;
};
PinnedPool g_pinned;

struct ChunkIn {
    size_t first = 0, count = 0;                       // range of the caller's lists
    std::vector<PinnedPool::Buf> raws;                 // per file of the range (p == nullptr: unreadable)
    std::vector<std::string> read_err;
    long long read_ms = 0;
};

struct ChunkOut {
    std::vector<size_t> idx;                           // files of the range that were read (offsets into the range)
    std::vector<uint8_t> tiles, labels;
    std::vector<int32_t> xy, start, cnt;
    long long device_ms = 0;
};

struct ChunkText {
    std::vector<std::string> con, err, lg;             // per read file: console, stderr, log text
    int ok = 0;
    long long art_ms = 0;
};

constexpr int kCapPoints = 1 << 15, kCapContours = 64;     // postprocess keeps components >= 6 % of the tile: <= 16

// I/O and artefact threads of directory mode: one per image of the chunk up to MEDSEG_IO_THREADS (default 16 -- a GPU's share of a
// host, never the whole machine: an 8-GPU node runs eight of these pools)
int io_threads_for(size_t n)
{
    static const int cap = std::max(1, env_int("MEDSEG_IO_THREADS", 16));
    return (int)std::max<size_t>(1, std::min<size_t>(n, (size_t)cap));
}

ChunkIn read_chunk(const std::vector<std::string> &paths, const std::vector<int> &widths, const std::vector<int> &heights,
                   size_t first, size_t count)
{
    ChunkIn in;
    in.first = first; in.count = count;
    in.raws.resize(count); in.read_err.resize(count);
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int nt = io_threads_for(count);
#pragma omp parallel for schedule(dynamic) num_threads(nt)
    for (long long k = 0; k < (long long)count; ++k) {
        try {
            const Preprocess::RawView view(paths[first + k], widths[first + k], heights[first + k]);
            in.raws[k] = g_pinned.acquire(view.samples());
            std::memcpy(in.raws[k].p, view.data(), view.samples() * sizeof(uint16_t));
        } catch (const std::exception &e) {
            in.read_err[k] = std::string("Processing error: ") + e.what() + " (" + paths[first + k] + ")";
            g_pinned.release(in.raws[k]);
            in.raws[k] = PinnedPool::Buf{};
        }
    }
    in.read_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return in;
}

ChunkOut device_chunk(const ChunkIn &in, const std::vector<int> &widths, const std::vector<int> &heights, mi_unet_group_t *group)
{
    ChunkOut out;
    std::vector<const uint16_t *> ptrs;
    std::vector<int> ws, hs;
    const int C = g_cfg.in_ch;
    for (size_t k = 0; k < in.count; ++k)
        if (in.read_err[k].empty()) {
            out.idx.push_back(k);
            for (int c = 0; c < C; ++c) {      // one plane per file: it feeds every input channel (include/mi_unet.h, mi_unet_infer_raw16)
                ptrs.push_back(in.raws[k].p); ws.push_back(widths[in.first + k]); hs.push_back(heights[in.first + k]);
            }
        }
    if (out.idx.empty()) return out;
    const size_t hw = (size_t)g_cfg.height * g_cfg.width, m = out.idx.size();
    std::vector<uint8_t> tiles_c(C > 1 ? hw * m * C : 0);
    out.tiles.resize(hw * m); out.labels.resize(hw * m);
    out.xy.resize(m * kCapPoints * 2); out.start.resize(m * (kCapContours + 1)); out.cnt.resize(m);
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (!group) throw std::runtime_error("Engine not initialized");
    if (mi_unet_group_segment_raw16(group, ptrs.data(), ws.data(), hs.data(), (int)m, C > 1 ? tiles_c.data() : out.tiles.data(),
                                    out.labels.data(), out.xy.data(), kCapPoints, out.start.data(), kCapContours,
                                    out.cnt.data()) != MI_UNET_OK)
        throw std::runtime_error(std::string("Inference failed: ") + mi_unet_last_error());
    if (C > 1)                                 // the artefact is the grey tile: channel 0 of the replicated planes
        for (size_t p = 0; p < hw * m; ++p) out.tiles[p] = tiles_c[p * C];
    out.device_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return out;
}

ChunkText artefact_chunk(const ChunkIn &in, const ChunkOut &out, const std::vector<std::string> &paths, const std::vector<int> &widths,
                         const std::vector<int> &heights, const std::string &output_dir)
{
    const size_t m = out.idx.size(), hw = (size_t)g_cfg.height * g_cfg.width;
    ChunkText tx;
    tx.con.resize(m); tx.err.resize(m); tx.lg.resize(m);
    std::vector<char> done(m, 0);
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int nt = io_threads_for(m);
#pragma omp parallel for schedule(dynamic) num_threads(nt)
    for (long long k = 0; k < (long long)m; ++k) {
        const size_t i = in.first + out.idx[k];
        std::ostringstream con, lg;
        medseg::set_png_threads(m > 1 ? 1 : 16);    // the images of a chunk are already written in parallel: no band threads inside
        try {
            const std::string base_name = fs::path(paths[i]).stem().string();
            lg << "\n=== Processing Image: " << fs::path(paths[i]).filename().string() << " ===" << std::endl;
            Image8 tile(g_cfg.height, g_cfg.width, 1), vis(g_cfg.height, g_cfg.width, 1);
            std::copy(out.tiles.begin() + k * hw, out.tiles.begin() + (k + 1) * hw, tile.data.begin());
            std::copy(out.labels.begin() + k * hw, out.labels.begin() + (k + 1) * hw, vis.data.begin());
            if (!Preprocess::write_preprocess_outputs(tile, paths[i], output_dir + "/" + base_name + "_normalized.png",
                                                      output_dir + "/" + base_name + "_original_sizes.json", widths[i], heights[i]))
                throw std::runtime_error("Preprocessing failed");
            if (!medseg::write_png(output_dir + "/" + base_name + "_mask.png", vis, /*level0=*/true))
                throw std::runtime_error("Failed to save mask");
            std::vector<medseg::Contour> contours;
            if (out.cnt[k] < 0) {                      // capacity overflow on the device: fall back to the host tracer
                contours = Mask2Polygon::extract_contours(vis);
            } else {
                const int32_t *st = &out.start[k * (kCapContours + 1)], *pts = &out.xy[k * (size_t)kCapPoints * 2];
                for (int c = 0; c < out.cnt[k]; ++c) {
                    medseg::Contour cc;
                    for (int q = st[c]; q < st[c + 1]; ++q) cc.emplace_back(pts[2 * q], pts[2 * q + 1]);
                    contours.push_back(std::move(cc));
                }
            }
            Mask2Polygon::write_polygon_outputs(contours, tile, output_dir, base_name, widths[i], heights[i], con);
            lg << "Processing completed for: " << base_name << std::endl;
            done[k] = 1;
        } catch (const std::exception &e) {
            tx.err[k] = std::string("Processing error: ") + e.what() + "\n";
            lg << "Processing error: " << e.what() << std::endl;
        }
        tx.con[k] = con.str(); tx.lg[k] = lg.str();
    }
    for (size_t k = 0; k < m; ++k) tx.ok += done[k];
    tx.art_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return tx;
}

// the all-device route of process_image_batch; returns the number of images that succeeded.
// Stages: read the files of chunk k+1 || device work of chunk k || PNG / JSON artefacts of chunk k-1.
// Two device lanes by default when the call has more than one piece (the second lane is a clone of the engine group: shared weights, own
// buffers / streams / worker threads), so that the exposed head of one piece's device call (upload + preprocess of its first images) and
// its tail (postprocess, contours, download of its last ones) run beside the other piece's network.  Round 3 measured no gain from it
// (1.77 vs 1.78 ms per image over 64 files: the file reads were the critical path then); with the reads out of the way -- MAP_POPULATE,
// host/preprocess.cpp -- same card, two rounds: 16 files 626 / 615 -> 647 / 652 images/s, 64 files 702 / 699 -> 804 / 719
// (profiles/r04_facade_chunks.txt).  MEDSEG_DEVICE_LANES=1 keeps one lane (half the activation memory).
int process_batch_pipelined(const std::vector<std::string> &paths, const std::vector<int> &widths, const std::vector<int> &heights,
                            const std::string &output_dir)
{
    auto &log_file = get_log_file();
    mi_unet_group_t *lanes[2] = { nullptr, nullptr };
    int n_lanes = 1;
    {
        std::lock_guard<std::mutex> lk(g_state_mutex);
        if (!g_group) throw std::runtime_error("Engine not initialized");
        lanes[0] = g_group;
        if (paths.size() >= 16 && env_int("MEDSEG_DEVICE_LANES", 2) >= 2) {      // (more than one piece: see `step` below)
            if (!g_lane2 && mi_unet_group_clone(g_group, &g_lane2) != MI_UNET_OK) {
                if (log_file.is_open()) log_file << "Warning: second device lane unavailable (" << mi_unet_last_error() << ")" << std::endl;
                g_lane2 = nullptr;
            }
            if (g_lane2) { lanes[1] = g_lane2; n_lanes = 2; }
        }
    }
    // a chunk = one micro-batch on every device of the group ... unless the whole call fits into one: then it is cut into four
    // pieces (at least four images each), so that reading piece k + 1, the device work of k (two lanes: k and k + 1) and the
    // artefacts of k - 1 overlap inside a 16-file call too.  Same card, 16 files, one lane: one piece 568, two 596, four 534 images/s
    // (smaller network batches cost more than the overlap returns); two lanes: one piece 527, two 615-637, four 674
    // (profiles/r04_facade_chunks.txt; MEDSEG_PIPELINE_CHUNK overrides the piece size)
    const size_t n = paths.size(), full = (size_t)std::max(1, g_cfg.max_batch) * (size_t)std::max(1, mi_unet_group_size(lanes[0]));
    size_t step = full;
    if (n <= full) step = n_lanes == 2 ? std::max<size_t>(4, (n + 3) / 4) : std::max<size_t>(8, (n + 1) / 2);
    if (const int forced = env_int("MEDSEG_PIPELINE_CHUNK", 0); forced > 0) step = std::min<size_t>(full, (size_t)forced);
    int ok = 0;
    auto emit = [&](const ChunkIn &in, const ChunkOut &out, const ChunkText &tx) {
        for (size_t k = 0; k < tx.con.size(); ++k) {
            std::cout << tx.con[k] << std::flush;
            std::cerr << tx.err[k] << std::flush;
            if (log_file.is_open()) log_file << tx.lg[k] << std::flush;
        }
        if (log_file.is_open())
            log_file << "Batch read time: " << in.read_ms << " ms for " << in.count << " files; Batch device time: " << out.device_ms
                     << " ms for " << out.idx.size() << " images; Batch artefact time: " << tx.art_ms << " ms" << std::endl;
        ok += tx.ok;
    };
    struct Stage { ChunkIn in; ChunkOut out; std::string dev_err; };
    struct InFlight { std::shared_ptr<Stage> st; std::future<void> done; };
    std::vector<InFlight> dev_q;                       // device work in flight, oldest first, at most n_lanes entries
    std::future<ChunkIn> next_read = std::async(std::launch::async, read_chunk, std::cref(paths), std::cref(widths), std::cref(heights),
                                                (size_t)0, std::min(step, n));
    std::future<ChunkText> pending_art;
    std::shared_ptr<Stage> art_stage;                  // keeps the chunk alive while its artefacts are being written
    auto retire_oldest = [&]() {                       // device work of the oldest chunk is over: hand it to the artefact stage
        InFlight f = std::move(dev_q.front());
        dev_q.erase(dev_q.begin());
        f.done.get();
        std::shared_ptr<Stage> st = f.st;
        if (!st->dev_err.empty()) {
            // this chunk's images fail (message as process_single_image's); chunks already done keep their successes and
            // the chunks behind it still run
            const std::string msg = "Processing error: " + st->dev_err + " (files " + std::to_string(st->in.first) + ".." +
                                    std::to_string(st->in.first + st->in.count - 1) + " of the batch)";
            std::cerr << msg << std::endl;
            if (log_file.is_open()) log_file << msg << std::endl;
            return;
        }
        if (pending_art.valid()) emit(art_stage->in, art_stage->out, pending_art.get());
        art_stage = st;
        pending_art = std::async(std::launch::async, [st, &paths, &widths, &heights, &output_dir] {
            return artefact_chunk(st->in, st->out, paths, widths, heights, output_dir);
        });
    };
    size_t k = 0;
    for (size_t first = 0; first < n; first += step, ++k) {
        auto st = std::make_shared<Stage>();
        st->in = next_read.get();
        if (first + step < n)
            next_read = std::async(std::launch::async, read_chunk, std::cref(paths), std::cref(widths), std::cref(heights),
                                   first + step, std::min(step, n - first - step));
        for (size_t q = 0; q < st->in.count; ++q)
            if (!st->in.read_err[q].empty()) {
                std::cerr << st->in.read_err[q] << std::endl;
                if (log_file.is_open()) log_file << st->in.read_err[q] << std::endl;
            }
        if ((int)dev_q.size() == n_lanes) retire_oldest();          // frees the lane this chunk will use (FIFO: chunk k - n_lanes)
        mi_unet_group_t *lane = lanes[k % n_lanes];
        dev_q.push_back({ st, std::async(std::launch::async, [st, lane, &widths, &heights] {
            try {
                st->out = device_chunk(st->in, widths, heights, lane);
            } catch (const std::exception &e) {
                st->dev_err = e.what();
            }
            for (auto &r : st->in.raws) { g_pinned.release(r); r = PinnedPool::Buf{}; }     // the RAW images are on the device's side now
        }) });
    }
    while (!dev_q.empty()) retire_oldest();
    if (pending_art.valid()) emit(art_stage->in, art_stage->out, pending_art.get());
    return ok;
}

}  // namespace

// Device-first form of the pipeline for N images at once (the reference loops files one by one, src/main.cpp:148-164).
// All-device route (default): the chunked three-stage pipeline above.  With MEDSEG_HOST_POSTPROCESS / _CONTOURS = 1:
// RAW16 -> [device: min/max, bilinear resample, quantise, UNet, argmax] -> per image on the host: PNG/JSON artefacts,
// postprocess_mask, contours.  Returns the number of images that succeeded.
int process_image_batch(const std::vector<std::string> &raw_paths, const std::vector<int> &widths,
                        const std::vector<int> &heights, const std::string &output_dir)
{
    auto &log_file = get_log_file();
    int ok = 0;
    try {
        mi_unet_group_t *group = get_engine_group();
        if (!group) throw std::runtime_error("Engine not initialized");
        const size_t n = raw_paths.size();
        if (widths.size() != n || heights.size() != n) throw std::runtime_error("widths/heights do not match raw_paths");
        if (n > 0 && device_postprocess_requested() && device_contours_requested()) {
            std::lock_guard<std::mutex> lk(g_batch_mutex);     // one directory-mode call at a time: it owns both device lanes
            return process_batch_pipelined(raw_paths, widths, heights, output_dir);
        }
        std::vector<std::vector<uint16_t>> raws(n);
        std::vector<const uint16_t *> ptrs;
        std::vector<int> ws, hs;
        std::vector<size_t> idx;                           // images that could be read
        std::vector<std::string> read_err(n);
        const auto t_read = std::chrono::high_resolution_clock::now();
        const int io_threads = io_threads_for(n);
#pragma omp parallel for schedule(dynamic) num_threads(io_threads)   // independent file reads; messages in file order below
        for (long long i = 0; i < (long long)n; ++i) {
            try {
                raws[i] = Preprocess::read_raw16(raw_paths[i], widths[i], heights[i]);
            } catch (const std::exception &e) {
                read_err[i] = std::string("Processing error: ") + e.what() + " (" + raw_paths[i] + ")";
                raws[i].clear();
            }
        }
        const int C = g_cfg.in_ch;
        for (size_t i = 0; i < n; ++i) {
            if (read_err[i].empty()) {
                for (int c = 0; c < C; ++c) { ptrs.push_back(raws[i].data()); ws.push_back(widths[i]); hs.push_back(heights[i]); }
                idx.push_back(i);
            } else {
                std::cerr << read_err[i] << std::endl;
                if (log_file.is_open()) log_file << read_err[i] << std::endl;
            }
        }
        if (log_file.is_open())
            log_file << "Batch read time: " << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t_read).count()
                     << " ms for " << n << " files" << std::endl;
        const size_t hw = (size_t)g_cfg.height * g_cfg.width;
        std::vector<uint8_t> tiles(hw * idx.size() * C), labels(hw * idx.size());
        const auto t0 = std::chrono::high_resolution_clock::now();
        const bool dev_post = device_postprocess_requested();
        {
            std::lock_guard<std::mutex> lk(g_batch_mutex);
            mi_unet_group_set_postprocess(group, dev_post ? 1 : 0);
            const int rc = idx.empty() ? MI_UNET_OK : mi_unet_group_infer_raw16(group, ptrs.data(), ws.data(), hs.data(), (int)idx.size(),
                                                                                tiles.data(), labels.data(), nullptr);
            mi_unet_group_set_postprocess(group, 0);
            if (rc != MI_UNET_OK) throw std::runtime_error(std::string("Inference failed: ") + mi_unet_last_error());
        }
        if (C > 1) {                               // keep channel 0 (the planes are replicas): the grey artefact tile
            for (size_t p = 0; p < hw * idx.size(); ++p) tiles[p] = tiles[p * C];
            tiles.resize(hw * idx.size());
        }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        if (log_file.is_open()) log_file << "Batch inference time: " << ms << " ms for " << idx.size() << " images" << std::endl;
        for (size_t k = 0; k < idx.size(); ++k) {
            const size_t i = idx[k];
            try {
                if (log_file.is_open())
                    log_file << "\n=== Processing Image: " << fs::path(raw_paths[i]).filename().string() << " ===" << std::endl;
                Image8 tile(g_cfg.height, g_cfg.width, 1), mask(g_cfg.height, g_cfg.width, 1);
                std::copy(tiles.begin() + k * hw, tiles.begin() + (k + 1) * hw, tile.data.begin());
                std::copy(labels.begin() + k * hw, labels.begin() + (k + 1) * hw, mask.data.begin());
                finish_image(raw_paths[i], widths[i], heights[i], output_dir, tile, std::move(mask), dev_post);
                if (log_file.is_open()) log_file << "Processing completed for: " << fs::path(raw_paths[i]).stem().string() << std::endl;
                ++ok;
            } catch (const std::exception &e) {
                std::cerr << "Processing error: " << e.what() << std::endl;
                if (log_file.is_open()) log_file << "Processing error: " << e.what() << std::endl;
            }
        }
    } catch (const std::exception &e) {
        std::cerr << "Processing error: " << e.what() << std::endl;
        if (log_file.is_open()) log_file << "Processing error: " << e.what() << std::endl;
    }
    return ok;
}

// One image on the CALLING THREAD'S own context (the reference's thread_local TensorRTContext, src/process.cpp:15): callers
// on different threads run concurrently.  The image's log block is collected and written in one piece, so blocks of
// concurrent images do not interleave (the reference's global stream is written unguarded, src/initialize.cpp:22).
bool process_single_image(const std::string &raw_path, int width, int height, const std::string &output_dir)
{
    std::ostringstream lg;
    auto flush_log = [&lg] {
        std::lock_guard<std::mutex> lk(g_log_mutex);
        if (g_log_file.is_open()) g_log_file << lg.str() << std::flush;
    };
    try {
        mi_unet_t *ctx = get_thread_local_context();           // throws "Engine not initialized"
        lg << "\n=== Processing Image: " << fs::path(raw_path).filename().string() << " ===" << std::endl;
        const std::string base_name = fs::path(raw_path).stem().string();
        const auto total_start = std::chrono::high_resolution_clock::now();

        if (host_preprocess_requested()) {
            // the reference's own order: CPU preprocess -> PNG on disk -> read back -> inference (src/process.cpp:211-224)
            const std::string preprocessed_png_path = output_dir + "/" + base_name + "_normalized.png";
            const std::string size_json_path = output_dir + "/" + base_name + "_original_sizes.json";
            bool pre_ok;
            if (g_cfg.width == 512 && g_cfg.height == 512) {
                pre_ok = Preprocess::preprocess_raw(raw_path, preprocessed_png_path, size_json_path, width, height);
            } else {                                           // a tile size the reference never had: same arithmetic, engine's size
                try {
                    const std::vector<uint16_t> raw = Preprocess::read_raw16(raw_path, width, height);
                    pre_ok = Preprocess::write_preprocess_outputs(Preprocess::resample_normalize(raw.data(), width, height, g_cfg.width, g_cfg.height),
                                                                  raw_path, preprocessed_png_path, size_json_path, width, height);
                } catch (const std::exception &e) {
                    std::cerr << "preprocess_raw error: " << e.what() << '\n';
                    pre_ok = false;
                }
            }
            if (!pre_ok) throw std::runtime_error("Preprocessing failed");
            const Image8 gray_img = medseg::read_png(preprocessed_png_path, /*as_color=*/false);
            if (gray_img.empty()) throw std::runtime_error("Failed to read preprocessed image");
            const auto infer_start = std::chrono::high_resolution_clock::now();
            Image8 pred_mask = execute_inference(gray_img);
            const auto infer_ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                                      std::chrono::high_resolution_clock::now() - infer_start).count();
            lg << "Inference time: " << infer_ms << " ms" << std::endl;
            finish_image(raw_path, width, height, output_dir, gray_img, std::move(pred_mask), false);
        } else if (device_postprocess_requested() && device_contours_requested()) {
            // all-device route: RAW16 -> tile -> UNet -> postprocess_mask -> mask_to_image -> contours in ONE call on this
            // thread's context; the mapped file is copied once, into pinned staging; the five artefacts
            // are written concurrently.  Per-stage times follow the reference's two log lines.
            using clk = std::chrono::steady_clock;
            auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
            const auto t_read = clk::now();
            std::unique_ptr<Preprocess::RawView> raw;
            try {
                raw.reset(new Preprocess::RawView(raw_path, width, height));
            } catch (const std::exception &e) {
                std::cerr << "preprocess_raw error: " << e.what() << '\n';
                throw std::runtime_error("Preprocessing failed");
            }
            const double read_ms = ms_since(t_read);
            const int C = g_cfg.in_ch;
            const std::vector<const uint16_t *> planes(C, raw->data());   // one plane feeds every input channel
            const std::vector<int> ws(C, width), hs(C, height);
            const size_t hw = (size_t)g_cfg.height * g_cfg.width;
            std::vector<uint8_t> tile_c(C > 1 ? hw * C : 0);
            Image8 tile(g_cfg.height, g_cfg.width, 1), vis(g_cfg.height, g_cfg.width, 1);
            std::vector<int32_t> xy((size_t)kCapPoints * 2), start(kCapContours + 1);
            int32_t cnt = 0;
            const auto infer_start = clk::now();
            const int rc = mi_unet_segment_raw16(ctx, planes.data(), ws.data(), hs.data(), 1, C > 1 ? tile_c.data() : tile.data.data(),
                                                 vis.data.data(), xy.data(), kCapPoints, start.data(), kCapContours, &cnt);
            if (rc != MI_UNET_OK) throw std::runtime_error(std::string("Inference failed: ") + mi_unet_last_error());
            raw.reset();
            if (C > 1)
                for (size_t p = 0; p < hw; ++p) tile.data[p] = tile_c[p * C];
            const double device_ms = ms_since(infer_start);
            float st[MI_UNET_N_STAGES] = {};
            mi_unet_last_stage_ms(ctx, st);
            lg << "Inference time: " << (long long)device_ms << " ms" << std::endl;
            std::vector<medseg::Contour> contours;
            if (cnt < 0) {                             // capacity overflow on the device: the host tracer takes over
                contours = Mask2Polygon::extract_contours(vis);
            } else {
                for (int k = 0; k < cnt; ++k) {
                    medseg::Contour cc;
                    for (int q = start[k]; q < start[k + 1]; ++q) cc.emplace_back(xy[2 * q], xy[2 * q + 1]);
                    contours.push_back(std::move(cc));
                }
            }
            // artefacts: {normalized.png + sizes.json} || {mask.png} || {overlay.png + polygon json}
            const auto t_art = clk::now();
            double norm_ms = 0, mask_ms = 0, poly_ms = 0;
            std::ostringstream con;
            auto f_norm = std::async(std::launch::async, [&] {
                const auto t0 = clk::now();
                const bool ok = Preprocess::write_preprocess_outputs(tile, raw_path, output_dir + "/" + base_name + "_normalized.png",
                                                                     output_dir + "/" + base_name + "_original_sizes.json", width, height);
                norm_ms = ms_since(t0);
                return ok;
            });
            auto f_mask = std::async(std::launch::async, [&] {
                const auto t0 = clk::now();
                const bool ok = medseg::write_png(output_dir + "/" + base_name + "_mask.png", vis, /*level0=*/true);
                mask_ms = ms_since(t0);
                return ok;
            });
            {
                const auto t0 = clk::now();
                Mask2Polygon::write_polygon_outputs(contours, tile, output_dir, base_name, width, height, con);
                poly_ms = ms_since(t0);
            }
            const bool norm_ok = f_norm.get(), mask_ok = f_mask.get();
            std::cout << con.str() << std::flush;
            if (!norm_ok) throw std::runtime_error("Preprocessing failed");
            if (!mask_ok) throw std::runtime_error("Failed to save mask");
            char line[512];
            std::snprintf(line, sizeof line,
                          "  Stage times (ms): read %.2f | device call %.2f = upload+preprocess %.2f, network %.2f, postprocess %.2f, "
                          "contours %.2f, download %.2f | artefacts %.2f = normalized.png+sizes.json %.2f || mask.png %.2f || "
                          "overlay.png+polygon.json %.2f",
                          read_ms, device_ms, st[MI_UNET_STAGE_UPLOAD_PRE], st[MI_UNET_STAGE_NETWORK], st[MI_UNET_STAGE_POSTPROCESS],
                          st[MI_UNET_STAGE_CONTOURS], st[MI_UNET_STAGE_DOWNLOAD], ms_since(t_art), norm_ms, mask_ms, poly_ms);
            lg << line << std::endl;
        } else {
            // device-first with a host tail (MEDSEG_HOST_POSTPROCESS / MEDSEG_HOST_CONTOURS = 1): min/max + resample + quantise
            // run on the GPU in front of the network; the tile comes back once, for the _normalized.png artefact
            std::vector<uint16_t> raw;
            try {
                raw = Preprocess::read_raw16(raw_path, width, height);
            } catch (const std::exception &e) {
                std::cerr << "preprocess_raw error: " << e.what() << '\n';
                throw std::runtime_error("Preprocessing failed");
            }
            const int C = g_cfg.in_ch;
            const std::vector<const uint16_t *> planes(C, raw.data());   // one plane feeds every input channel
            const std::vector<int> ws(C, width), hs(C, height);
            const size_t hw = (size_t)g_cfg.height * g_cfg.width;
            std::vector<uint8_t> tile_c(hw * C);
            Image8 tile(g_cfg.height, g_cfg.width, 1), pred_mask(g_cfg.height, g_cfg.width, 1);
            const auto infer_start = std::chrono::high_resolution_clock::now();
            const bool dev_post = device_postprocess_requested();
            mi_unet_set_postprocess(ctx, dev_post ? 1 : 0);    // the context belongs to this thread: no lock
            const int rc = mi_unet_infer_raw16(ctx, planes.data(), ws.data(), hs.data(), 1, tile_c.data(), pred_mask.data.data(), nullptr);
            mi_unet_set_postprocess(ctx, 0);
            if (rc != MI_UNET_OK) throw std::runtime_error(std::string("Inference failed: ") + mi_unet_last_error());
            for (size_t p = 0; p < hw; ++p) tile.data[p] = tile_c[p * C];
            const auto infer_ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                                      std::chrono::high_resolution_clock::now() - infer_start).count();
            lg << "Inference time: " << infer_ms << " ms" << std::endl;
            finish_image(raw_path, width, height, output_dir, tile, std::move(pred_mask), dev_post);
        }

        const auto total_ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                                  std::chrono::high_resolution_clock::now() - total_start).count();
        lg << "Total processing time: " << total_ms << " ms" << std::endl;
        lg << "Processing completed for: " << base_name << std::endl;
        flush_log();
        std::cout << "Total processing time: " << total_ms << " ms" << std::endl;
        return true;
    } catch (const std::exception &e) {
        std::cerr << "Processing error: " << e.what() << std::endl;
        lg << "Processing error: " << e.what() << std::endl;
        flush_log();
        return false;
    }
}

// Releases the calling thread's context (as the reference does, src/cleanup.cpp:16-35), the engine group and the log.
// Contexts of other threads notice the generation change and are released on their next use or when their thread ends
// (they share the weight blob, which lives until the last of them is gone).
void cleanup_resources()
{
    try {
        std::lock_guard<std::mutex> lk(g_state_mutex);
        auto &log_file = g_log_file;
        if (log_file.is_open()) log_file << "\n=== Cleaning Up Resources ===" << std::endl;
        if (t_context.h) {
            t_context.release();
            if (log_file.is_open()) log_file << "Execution context destroyed" << std::endl;
        }
        ++g_generation;
        g_pinned.clear();
        if (g_lane2) { mi_unet_group_destroy(g_lane2); g_lane2 = nullptr; }
        if (g_group) {
            mi_unet_group_destroy(g_group);         // every device's buffers, streams, worker thread, weights
            g_group = nullptr;
            if (log_file.is_open()) log_file << "MI355X UNet engine destroyed" << std::endl;
        }
        if (log_file.is_open()) {
            log_file << "All resources cleaned up successfully" << std::endl;
            log_file.close();
        }
        std::cout << "Resources cleaned up successfully" << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "Cleanup error: " << e.what() << std::endl;
    }
}

}  // namespace MedicalSeg
