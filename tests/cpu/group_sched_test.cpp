// CPU test of the multi-device group's scheduling (csrc/group_sched.h) with a stub backend: every "device" is a function
// that records which items it was given.  Prints one line per check; exit code 0 = all passed.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <numeric>

#include "../../unet-medical-image-contour-segmentation-cpp_amd/csrc/group_sched.h"

using namespace miunet;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

int main()
{
    // 1. a batch of B items over R ranks: every item is processed exactly once, by the rank that owns it, into ITS slice
    for (int R : { 1, 2, 3, 8 })
        for (int B : { 0, 1, 5, 16, 37, 512 }) {
            std::vector<std::unique_ptr<Worker>> workers;
            for (int r = 0; r < R; ++r) workers.emplace_back(new Worker());
            std::vector<int> owner(B, -1), hits(B, 0);
            std::vector<int> in(B), out(B, 0);
            std::iota(in.begin(), in.end(), 1000);
            int bad = -2;
            std::string msg;
            const int rc = run_on_all_ranks(workers, [&](int r) {
                int lo, hi;
                shard_range(B, r, R, lo, hi);
                for (int i = lo; i < hi; ++i) { owner[i] = r; ++hits[i]; out[i] = 2 * in[i]; }      // the stub "engine"
                return 0;
            }, [] { return std::string(); }, bad, msg);
            CHECK(rc == 0 && bad == -1, "R=%d B=%d rc=%d", R, B, rc);
            int prev = 0;
            for (int i = 0; i < B; ++i) {
                CHECK(hits[i] == 1 && out[i] == 2 * in[i], "R=%d B=%d item %d hits=%d", R, B, i, hits[i]);
                CHECK(owner[i] >= prev, "R=%d B=%d item %d: owners not in rank order", R, B, i);
                prev = owner[i];
            }
            for (int r = 0, covered = 0; r < R; ++r) {
                int lo, hi;
                shard_range(B, r, R, lo, hi);
                CHECK(lo == covered && hi - lo >= B / R && hi - lo <= B / R + 1, "R=%d B=%d rank %d range [%d,%d)", R, B, r, lo, hi);
                covered = hi;
                if (r == R - 1) CHECK(hi == B, "R=%d B=%d last range ends at %d", R, B, hi);
            }
        }
    // 2. the ranks really run concurrently: 4 ranks that each sleep 100 ms finish in well under 400 ms
    {
        std::vector<std::unique_ptr<Worker>> workers;
        for (int r = 0; r < 4; ++r) workers.emplace_back(new Worker());
        int bad;
        std::string msg;
        const auto t0 = std::chrono::steady_clock::now();
        run_on_all_ranks(workers, [](int) { std::this_thread::sleep_for(std::chrono::milliseconds(100)); return 0; },
                         [] { return std::string(); }, bad, msg);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CHECK(ms < 300.0, "4 x 100 ms took %.0f ms: workers are not concurrent", ms);
    }
    // 3. a failing rank: its code and ITS thread's message come back; the lowest failing rank wins; workers stay usable
    {
        std::vector<std::unique_ptr<Worker>> workers;
        for (int r = 0; r < 5; ++r) workers.emplace_back(new Worker());
        thread_local std::string tls_err;
        int bad;
        std::string msg;
        const int rc = run_on_all_ranks(workers, [&](int r) {
            if (r == 2 || r == 4) { tls_err = "device " + std::to_string(r) + " fell over"; return 3 + r; }
            return 0;
        }, [&] { return tls_err; }, bad, msg);
        CHECK(rc == 5 && bad == 2 && msg == "device 2 fell over", "rc=%d bad=%d msg=%s", rc, bad, msg.c_str());
        std::atomic<int> n{ 0 };
        const int rc2 = run_on_all_ranks(workers, [&](int) { ++n; return 0; }, [] { return std::string(); }, bad, msg);
        CHECK(rc2 == 0 && n == 5, "second call rc=%d n=%d", rc2, (int)n);
    }
    std::printf("%s\n", failures ? "FAILED" : "all group scheduling checks passed");
    return failures ? 1 : 0;
}
