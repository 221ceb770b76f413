// cli.cpp -- the interactive front end of the reference (src/main.cpp:18-198) on top of the facade: same commands
// (init / process [-r] / exit / help), same extension filter, same output-directory mirroring and success/fail counters.
// Directory mode differs in ONE way: the files of a directory level are handed to MedicalSeg::process_image_batch, so
// the device sees one batch instead of a loop of single images (set MEDSEG_CLI_SINGLE=1 for the reference's loop).
#include <algorithm>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/medseg/cleanup.h"
#include "../../include/medseg/initialize.h"
#include "../../include/medseg/process.h"

namespace fs = std::filesystem;
using namespace MedicalSeg;

namespace {

bool is_16bit_image(const std::string &path)
{
    static const char *exts[] = { ".raw", ".dcm", ".tif", ".tiff" };        // src/main.cpp:19-21
    std::string ext = fs::path(path).extension().string();
    std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
    return std::find(std::begin(exts), std::end(exts), ext) != std::end(exts);
}

std::vector<std::string> find_16bit_images(const std::string &dir, bool recursive)
{
    std::vector<std::string> out;
    try {
        auto take = [&](const fs::directory_entry &e) {
            if (e.is_regular_file() && is_16bit_image(e.path().string())) out.push_back(e.path().string());
        };
        if (recursive) for (const auto &e : fs::recursive_directory_iterator(dir)) take(e);
        else for (const auto &e : fs::directory_iterator(dir)) take(e);
    } catch (const fs::filesystem_error &e) {
        std::cerr << "Directory error: " << e.what() << std::endl;
    }
    std::sort(out.begin(), out.end());
    return out;
}

void print_usage()
{
    std::cout << "\nMedical Image Segmentation Tool (MI355X)" << std::endl;
    std::cout << "Commands:" << std::endl;
    std::cout << "  init <weight_file>            - Initialize the UNet engine" << std::endl;
    std::cout << "  process [-r] <input> <width> <height> [output_dir] - Process file/directory" << std::endl;
    std::cout << "  exit                          - Cleanup and exit" << std::endl;
    std::cout << "\nOptions:" << std::endl;
    std::cout << "  -r                            - Recursively process directory" << std::endl;
    std::cout << "  <input>                       - Path to image file or directory" << std::endl;
}

}  // namespace

int main()
{
    bool initialized = false;
    std::string command;
    std::cout << "Welcome to Medical Image Segmentation Tool" << std::endl;
    print_usage();
    while (true) {
        std::cout << "\n> " << std::flush;
        if (!std::getline(std::cin, command)) command = "exit";              // EOF behaves like exit
        std::istringstream iss(command);
        std::string cmd;
        iss >> cmd;
        if (cmd == "init") {
            std::string path;
            iss >> path;
            if (path.empty()) { std::cerr << "Error: Missing weight file path" << std::endl; continue; }
            const std::string log_dir = fs::path(path).parent_path().string() + "/../log";     // src/main.cpp:87
            if (initialize_engine(path, log_dir)) { std::cout << "Engine initialized successfully" << std::endl; initialized = true; }
            else std::cerr << "Engine initialization failed" << std::endl;
        } else if (cmd == "process") {
            if (!initialized) { std::cerr << "Error: Engine not initialized" << std::endl; continue; }
            bool recursive = false;
            std::string input_path, output_dir, arg;
            int width = 0, height = 0;
            iss >> arg;
            if (arg == "-r") { recursive = true; iss >> input_path; } else input_path = arg;
            iss >> width >> height;
            if (input_path.empty() || !iss) { std::cerr << "Error: Invalid process command" << std::endl; continue; }
            iss >> output_dir;
            if (output_dir.empty()) output_dir = fs::path(input_path).parent_path().string();
            try {
                fs::create_directories(output_dir);
                if (fs::is_directory(input_path)) {
                    std::cout << "Processing directory: " << input_path << std::endl;
                    std::cout << "Recursive: " << (recursive ? "Yes" : "No") << std::endl;
                    const auto files = find_16bit_images(input_path, recursive);
                    if (files.empty()) { std::cerr << "No 16-bit images found in directory" << std::endl; continue; }
                    std::cout << "Found " << files.size() << " images to process" << std::endl;
                    int success_count = 0, fail_count = 0;
                    const char *single = std::getenv("MEDSEG_CLI_SINGLE");
                    // group by output sub-directory (the directory structure is mirrored in recursive mode, :150-156)
                    std::map<std::string, std::vector<std::string>> groups;
                    for (const auto &file : files) {
                        std::string file_output_dir = output_dir;
                        if (recursive) {
                            file_output_dir = (fs::path(output_dir) / fs::relative(file, input_path).parent_path()).string();
                            fs::create_directories(file_output_dir);
                        }
                        groups[file_output_dir].push_back(file);
                    }
                    for (const auto &g : groups) {
                        if (single && single[0] == '1') {
                            for (const auto &file : g.second) {
                                std::cout << "\nProcessing: " << file << std::endl;
                                if (process_single_image(file, width, height, g.first)) ++success_count; else ++fail_count;
                            }
                        } else {
                            for (const auto &file : g.second) std::cout << "\nProcessing: " << file << std::endl;
                            const int ok = process_image_batch(g.second, std::vector<int>(g.second.size(), width),
                                                               std::vector<int>(g.second.size(), height), g.first);
                            success_count += ok;
                            fail_count += (int)g.second.size() - ok;
                        }
                    }
                    std::cout << "\nDirectory processing completed:" << std::endl;
                    std::cout << "  Success: " << success_count << " files" << std::endl;
                    std::cout << "  Failed: " << fail_count << " files" << std::endl;
                } else if (fs::is_regular_file(input_path)) {
                    std::cout << "Processing file: " << input_path << std::endl;
                    if (process_single_image(input_path, width, height, output_dir)) std::cout << "Processing completed" << std::endl;
                    else std::cerr << "Processing failed" << std::endl;
                } else {
                    std::cerr << "Error: Input path is not a valid file or directory" << std::endl;
                }
            } catch (const std::exception &e) {
                std::cerr << "Processing error: " << e.what() << std::endl;
            }
        } else if (cmd == "exit") {
            if (initialized) cleanup_resources();
            std::cout << "Exiting..." << std::endl;
            break;
        } else if (cmd == "help") {
            print_usage();
        } else if (!cmd.empty()) {
            std::cerr << "Unknown command: " << cmd << std::endl;
        }
    }
    return 0;
}
