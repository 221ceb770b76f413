// lifecycle.cpp -- MedicalSeg::initialize_engine / process_single_image / cleanup_resources on top of the C-ABI
// (include/mi_unet.h).  Reference: src/initialize.cpp:26-91, src/process.cpp:123-262, src/cleanup.cpp:10-64.
// Same log file name, banner lines, message prefixes and bool/void error conventions; the TensorRT engine, the
// thread-local execution context and the CUDA graph are replaced by one mi_unet handle.
#include <chrono>
#include <filesystem>
#include <iostream>
#include <mutex>
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
mi_unet_t *g_engine = nullptr;
mi_unet_config g_cfg{};
std::ofstream g_log_file;
std::string g_log_path;
std::mutex g_infer_mutex;          // a handle serves one caller at a time (the reference's static staging vectors raced)
}  // namespace

bool initialize_engine(const std::string &trt_cache_path, const std::string &log_dir)
{
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
        if (g_engine) { mi_unet_destroy(g_engine); g_engine = nullptr; }
        mi_unet_default_config(&g_cfg);            // 512x512x1, 3 classes (src/process.cpp:70, :162)
        g_cfg.max_batch = 16;
        if (mi_unet_create(&g_cfg, &g_engine) != MI_UNET_OK || mi_unet_load_weights(g_engine, trt_cache_path.c_str()) != MI_UNET_OK) {
            g_log_file << "Error: Failed to initialize MI355X UNet engine: " << mi_unet_last_error() << std::endl;
            std::cerr << "Initialization error: " << mi_unet_last_error() << std::endl;
            if (g_engine) { mi_unet_destroy(g_engine); g_engine = nullptr; }
            return false;
        }
        g_log_file << "MI355X UNet engine initialized successfully" << std::endl;
        g_log_file << "  Input size: " << (size_t)g_cfg.height * g_cfg.width * g_cfg.in_ch << " bytes (u8)" << std::endl;
        g_log_file << "  Output size: " << (size_t)g_cfg.height * g_cfg.width << " bytes (classes=" << g_cfg.classes << ")" << std::endl;
        return true;
    } catch (const std::exception &e) {
        std::cerr << "Initialization error: " << e.what() << std::endl;
        if (g_log_file.is_open()) g_log_file << "Initialization error: " << e.what() << std::endl;
        return false;
    }
}

mi_unet_t *get_engine() { return g_engine; }
std::ofstream &get_log_file() { return g_log_file; }
std::string get_log_path() { return g_log_path; }

std::vector<Image8> execute_inference_batch(const std::vector<Image8> &gray_imgs)
{
    try {
        if (!g_engine) throw std::runtime_error("Engine not initialized");
        const size_t hw = (size_t)g_cfg.height * g_cfg.width;
        std::vector<uint8_t> in(hw * gray_imgs.size()), out(hw * gray_imgs.size());
        for (size_t i = 0; i < gray_imgs.size(); ++i) {
            const Image8 &g = gray_imgs[i];
            if (g.rows != g_cfg.height || g.cols != g_cfg.width || g.channels != 1)
                throw std::runtime_error("Input size must be 512x512 for fixed context");     // src/process.cpp:127
            std::copy(g.data.begin(), g.data.end(), in.begin() + i * hw);
        }
        {
            std::lock_guard<std::mutex> lk(g_infer_mutex);
            if (mi_unet_infer_u8(g_engine, in.data(), (int)gray_imgs.size(), out.data(), nullptr) != MI_UNET_OK)
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

Image8 execute_inference(const Image8 &gray_img)
{
    return execute_inference_batch({ gray_img })[0];
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

bool process_single_image(const std::string &raw_path, int width, int height, const std::string &output_dir)
{
    try {
        auto &log_file = get_log_file();
        if (!g_engine) throw std::runtime_error("Engine not initialized");
        log_file << "\n=== Processing Image: " << fs::path(raw_path).filename().string() << " ===" << std::endl;
        const std::string base_name = fs::path(raw_path).stem().string();
        const auto total_start = std::chrono::high_resolution_clock::now();

        const std::string preprocessed_png_path = output_dir + "/" + base_name + "_normalized.png";
        const std::string size_json_path = output_dir + "/" + base_name + "_original_sizes.json";
        const std::string pred_mask_path = output_dir + "/" + base_name + "_mask.png";

        if (!Preprocess::preprocess_raw(raw_path, preprocessed_png_path, size_json_path, width, height))
            throw std::runtime_error("Preprocessing failed");
        const Image8 gray_img = medseg::read_png(preprocessed_png_path, /*as_color=*/false);
        if (gray_img.empty()) throw std::runtime_error("Failed to read preprocessed image");

        const auto infer_start = std::chrono::high_resolution_clock::now();
        Image8 pred_mask = execute_inference(gray_img);
        const auto infer_ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                                  std::chrono::high_resolution_clock::now() - infer_start).count();
        log_file << "Inference time: " << infer_ms << " ms" << std::endl;

        pred_mask = postprocess_mask(pred_mask);
        if (!medseg::write_png(pred_mask_path, mask_to_image(pred_mask), /*level0=*/true))
            throw std::runtime_error("Failed to save mask");
        Mask2Polygon::process_single_mask(pred_mask_path, output_dir, size_json_path, preprocessed_png_path, base_name);

        const auto total_ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                                  std::chrono::high_resolution_clock::now() - total_start).count();
        log_file << "Total processing time: " << total_ms << " ms" << std::endl;
        log_file << "Processing completed for: " << base_name << std::endl;
        std::cout << "Total processing time: " << total_ms << " ms" << std::endl;
        return true;
    } catch (const std::exception &e) {
        std::cerr << "Processing error: " << e.what() << std::endl;
        auto &log_file = get_log_file();
        if (log_file.is_open()) log_file << "Processing error: " << e.what() << std::endl;
        return false;
    }
}

void cleanup_resources()
{
    try {
        auto &log_file = get_log_file();
        if (log_file.is_open()) log_file << "\n=== Cleaning Up Resources ===" << std::endl;
        if (g_engine) {
            mi_unet_destroy(g_engine);         // device buffers, stream, weights
            g_engine = nullptr;
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
