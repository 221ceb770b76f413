// cli.cpp -- line-oriented front end of the facade.  It speaks the reference REPL's command grammar and prints its
// messages (the contract of /root/reference/src/main.cpp: `init <file>`, `process [-r] <input> <w> <h> [outdir]`, `exit`,
// `help`), but is organised around the batch facade: a command table, a planner that turns an input path into
// (file, output directory) jobs, and a runner that hands every output directory's jobs to the device as ONE batch
// (MedicalSeg::process_image_batch; MEDSEG_CLI_SINGLE=1 runs the jobs one image per call instead).
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <filesystem>
#include <functional>
#include <iostream>
#include <iterator>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/medseg/cleanup.h"
#include "../../include/medseg/initialize.h"
#include "../../include/medseg/process.h"

namespace fs = std::filesystem;

namespace {

using Words = std::vector<std::string>;

// ---------------------------------------------------------------------------------------------------------------- planning
struct Job {
    fs::path file;       // a RAW16 input
    fs::path out_dir;    // where its five artefacts go
};

// The reference accepts these suffixes (case-insensitive) and reads every one of them as headerless RAW.
bool has_raw16_suffix(const fs::path &p)
{
    std::string s = p.extension().string();
    for (char &c : s) c = (char)std::tolower((unsigned char)c);
    return s == ".raw" || s == ".dcm" || s == ".tif" || s == ".tiff";
}

// Jobs for every matching regular file below `root` (one level, or the whole tree with `deep`); with `deep` the output tree
// mirrors the input tree.  Sorted by (output directory, file) so that a directory's files form one contiguous batch and the
// run order does not depend on the file system's enumeration order.
std::vector<Job> plan_directory(const fs::path &root, bool deep, const fs::path &out_root)
{
    std::vector<Job> jobs;
    auto consider = [&](const fs::directory_entry &e) {
        if (!e.is_regular_file() || !has_raw16_suffix(e.path())) return;
        const fs::path sub = deep ? fs::relative(e.path(), root).parent_path() : fs::path();
        jobs.push_back({ e.path(), sub.empty() ? out_root : out_root / sub });
    };
    try {
        if (deep) { for (const auto &e : fs::recursive_directory_iterator(root)) consider(e); }
        else { for (const auto &e : fs::directory_iterator(root)) consider(e); }
    } catch (const fs::filesystem_error &err) {
        std::cerr << "Directory error: " << err.what() << std::endl;
    }
    std::sort(jobs.begin(), jobs.end(), [](const Job &a, const Job &b) {
        return a.out_dir != b.out_dir ? a.out_dir < b.out_dir : a.file < b.file;
    });
    return jobs;
}

// ---------------------------------------------------------------------------------------------------------------- running
struct Tally {
    int succeeded = 0, failed = 0;
    void add(int ok, int total) { succeeded += ok; failed += total - ok; }
};

bool one_image_per_call()
{
    const char *v = std::getenv("MEDSEG_CLI_SINGLE");
    return v != nullptr && v[0] == '1';
}

// [first, last) share one output directory
void run_group(std::vector<Job>::const_iterator first, std::vector<Job>::const_iterator last, int width, int height, Tally &tally)
{
    const std::string out_dir = first->out_dir.string();
    fs::create_directories(first->out_dir);
    const int n = (int)std::distance(first, last);
    if (one_image_per_call()) {
        for (auto j = first; j != last; ++j) {
            std::cout << "\nProcessing: " << j->file.string() << std::endl;
            tally.add(MedicalSeg::process_single_image(j->file.string(), width, height, out_dir) ? 1 : 0, 1);
        }
        return;
    }
    std::vector<std::string> files;
    for (auto j = first; j != last; ++j) {
        std::cout << "\nProcessing: " << j->file.string() << std::endl;
        files.push_back(j->file.string());
    }
    tally.add(MedicalSeg::process_image_batch(files, std::vector<int>(n, width), std::vector<int>(n, height), out_dir), n);
}

void run_directory(const fs::path &input, bool deep, int width, int height, const fs::path &out_root)
{
    std::cout << "Processing directory: " << input.string() << std::endl;
    std::cout << "Recursive: " << (deep ? "Yes" : "No") << std::endl;
    const std::vector<Job> jobs = plan_directory(input, deep, out_root);
    if (jobs.empty()) {
        std::cerr << "No 16-bit images found in directory" << std::endl;
        return;
    }
    std::cout << "Found " << jobs.size() << " images to process" << std::endl;
    Tally tally;
    for (auto g = jobs.begin(); g != jobs.end();) {
        auto e = std::find_if(g, jobs.end(), [&](const Job &j) { return j.out_dir != g->out_dir; });
        run_group(g, e, width, height, tally);
        g = e;
    }
    std::cout << "\nDirectory processing completed:" << std::endl;
    std::cout << "  Success: " << tally.succeeded << " files" << std::endl;
    std::cout << "  Failed: " << tally.failed << " files" << std::endl;
}

void run_file(const fs::path &input, int width, int height, const fs::path &out_root)
{
    std::cout << "Processing file: " << input.string() << std::endl;
    if (MedicalSeg::process_single_image(input.string(), width, height, out_root.string())) std::cout << "Processing completed" << std::endl;
    else std::cerr << "Processing failed" << std::endl;
}

// ---------------------------------------------------------------------------------------------------------------- the shell
bool to_int(const std::string &s, int &v)
{
    try {
        size_t used = 0;
        v = std::stoi(s, &used);
        return used == s.size();
    } catch (const std::exception &) {
        return false;
    }
}

class Shell {
public:
    Shell()
    {
        table_["init"] = [this](const Words &w) { cmd_init(w); return true; };
        table_["process"] = [this](const Words &w) { cmd_process(w); return true; };
        table_["help"] = [](const Words &) { banner(); return true; };
        table_["exit"] = [this](const Words &) { cmd_exit(); return false; };
    }

    static void banner()
    {
        static const char *const lines[] = {
            "",
            "Medical Image Segmentation Tool (MI355X)",
            "Commands:",
            "  init <weight_file>            - Initialize the UNet engine",
            "  process [-r] <input> <width> <height> [output_dir] - Process file/directory",
            "  exit                          - Cleanup and exit",
            "",
            "Options:",
            "  -r                            - Recursively process directory",
            "  <input>                       - Path to image file or directory",
        };
        for (const char *l : lines) std::cout << l << std::endl;
    }

    // false = leave the loop
    bool dispatch(const std::string &line)
    {
        std::istringstream in(line);
        const Words w{ std::istream_iterator<std::string>(in), std::istream_iterator<std::string>() };
        if (w.empty()) return true;
        const auto it = table_.find(w[0]);
        if (it == table_.end()) {
            std::cerr << "Unknown command: " << w[0] << std::endl;
            return true;
        }
        return it->second(w);
    }

private:
    void cmd_init(const Words &w)
    {
        if (w.size() < 2) {
            std::cerr << "Error: Missing weight file path" << std::endl;
            return;
        }
        // the log directory sits next to the engine directory, as in the reference: <dir of the file>/../log
        const std::string log_dir = fs::path(w[1]).parent_path().string() + "/../log";
        // as the reference (src/main.cpp:88-93): the flag is only ever set, so after a failed RE-initialisation `process` still
        // reaches the facade (which reports "Engine not initialized" per image) and `exit` still runs cleanup_resources
        if (MedicalSeg::initialize_engine(w[1], log_dir)) {
            std::cout << "Engine initialized successfully" << std::endl;
            ready_ = true;
        } else {
            std::cerr << "Engine initialization failed" << std::endl;
        }
    }

    void cmd_process(const Words &w)
    {
        if (!ready_) {
            std::cerr << "Error: Engine not initialized" << std::endl;
            return;
        }
        size_t at = 1;
        const bool deep = at < w.size() && w[at] == "-r";
        if (deep) ++at;
        int width = 0, height = 0;
        if (at + 2 >= w.size() || !to_int(w[at + 1], width) || !to_int(w[at + 2], height)) {
            std::cerr << "Error: Invalid process command" << std::endl;
            return;
        }
        const fs::path input = w[at];
        const fs::path out_root = at + 3 < w.size() ? fs::path(w[at + 3]) : input.parent_path();
        try {
            fs::create_directories(out_root);
            if (fs::is_directory(input)) run_directory(input, deep, width, height, out_root);
            else if (fs::is_regular_file(input)) run_file(input, width, height, out_root);
            else std::cerr << "Error: Input path is not a valid file or directory" << std::endl;
        } catch (const std::exception &err) {
            std::cerr << "Processing error: " << err.what() << std::endl;
        }
    }

    void cmd_exit()
    {
        if (ready_) MedicalSeg::cleanup_resources();
        std::cout << "Exiting..." << std::endl;
    }

    std::unordered_map<std::string, std::function<bool(const Words &)>> table_;
    bool ready_ = false;
};

}  // namespace

int main()
{
    Shell shell;
    std::cout << "Welcome to Medical Image Segmentation Tool" << std::endl;
    Shell::banner();
    std::string line;
    for (;;) {
        std::cout << "\n> " << std::flush;
        if (!std::getline(std::cin, line)) line = "exit";            // end of input closes the session cleanly
        if (!shell.dispatch(line)) break;
    }
    return 0;
}
