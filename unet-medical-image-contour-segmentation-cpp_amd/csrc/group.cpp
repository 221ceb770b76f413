// group.cpp -- the multi-device group of include/mi_unet.h: one engine handle and one persistent host worker thread per
// device inside one process, contiguous image shards, weights packed once and sent device-to-device (RCCL broadcast over
// xGMI, or a peer copy), label maps gathered either by every rank's own D2H into the caller's buffer or by grouped
// ncclSend / ncclRecv into the first device.
//
// The reference has no counterpart: it is one image, one implicit device 0 (src/process.cpp:70; no cudaSetDevice anywhere);
// the slot this fills is its sequential file loop (src/main.cpp:148-164) and its per-thread context (src/process.cpp:15).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/mi_unet.h"
#include "engine_internal.h"
#include "group_sched.h"

using namespace miunet;

namespace {

// ---- RCCL, bound at run time: libmiunet.so has no load-time dependency on it (a single-device engine never needs it, and a
// host process may already carry its own copy, e.g. the one bundled with PyTorch).
struct Rccl {
    typedef void *comm_t;
    void *so = nullptr;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    static constexpr int kUint8 = 1;          // ncclUint8 / ncclChar = 0 is signed; rccl.h: ncclInt8 = 0, ncclUint8 = 1

    bool load(std::string &why)
    {
        if (so) return true;
        // MIUNET_RCCL_LIB: another library with the same eight entry points (tests/cpu/fake_rccl.cpp: the group's RCCL calls executed
        // on a one-GPU box, every transfer a device-to-device copy)
        const char *names[] = { getenv("MIUNET_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *n : names)
            if (n && *n && (so = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!so) { why = std::string("dlopen librccl: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(so, n); if (!p) why = std::string("librccl lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast || !Send || !Recv || !GetErrorString) {
            dlclose(so); so = nullptr;
            return false;
        }
        return true;
    }
};

}  // namespace

struct mi_unet_group {
    mi_unet_config cfg{};
    std::vector<int> devices;
    std::vector<mi_unet_t *> eng;
    std::vector<std::unique_ptr<Worker>> workers;
    Rccl rccl;
    std::vector<Rccl::comm_t> comms;          // one per rank when the devices are distinct and RCCL loaded, else empty
    std::string transport = "peer-copy";
    int gather = MI_UNET_GATHER_HOST;
    bool postprocess = false;
    // XGMI gather: per-rank device buffers for its shard (inputs, label maps), grown on demand; rank 0's holds the whole batch
    std::vector<uint8_t *> d_in, d_out;
    std::vector<size_t> cap_in, cap_out;
    uint8_t *h_all = nullptr;                 // pinned: rank 0's D2H target
    size_t cap_all = 0;
    std::mutex call_mutex;                    // one group call at a time
};

namespace {

// run fn(rank) on every rank's worker; first failure (by rank) becomes the caller's last error
int for_all_ranks(mi_unet_group *g, const std::function<int(int)> &fn)
{
    std::string msg;
    int bad_rank = -1;
    const int rc = run_on_all_ranks(g->workers, fn, [] { return std::string(mi_unet_last_error()); }, bad_rank, msg);
    if (rc) return engine_fail(rc, "rank " + std::to_string(bad_rank) + " (device " + std::to_string(g->devices[bad_rank]) + "): " + msg);
    return MI_UNET_OK;
}

int nccl_fail(mi_unet_group *g, int code, const char *what)
{
    return engine_fail(MI_UNET_EHIP, std::string(what) + ": " + g->rccl.GetErrorString(code));
}

#define HIP_TRY_G(expr)                                                                                        \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) return engine_fail(MI_UNET_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

int grow(uint8_t *&p, size_t &cap, size_t need, int device)
{
    if (need <= cap) return 0;
    HIP_TRY_G(hipSetDevice(device));
    if (p) HIP_TRY_G(hipFree(p));
    p = nullptr; cap = 0;
    HIP_TRY_G(hipMalloc(&p, need));
    cap = need;
    return 0;
}

}  // namespace

extern "C" {

int mi_unet_shard_range(int n_items, int rank, int world, int *lo, int *hi)
{
    if (!lo || !hi || n_items < 0 || world < 1 || rank < 0 || rank >= world) return engine_fail(MI_UNET_EARG, "mi_unet_shard_range: bad argument");
    shard_range(n_items, rank, world, *lo, *hi);
    return MI_UNET_OK;
}

int mi_unet_group_create(const mi_unet_config *cfg, const int *devices, int n_devices, mi_unet_group_t **out)
{
    if (!cfg || !out) return engine_fail(MI_UNET_EARG, "mi_unet_group_create: null argument");
    *out = nullptr;
    const int visible = mi_unet_device_count();
    if (visible <= 0) return engine_fail(MI_UNET_ENODEVICE, "no HIP device visible: libmiunet has no CPU fallback");
    std::vector<int> devs;
    if (devices) {
        if (n_devices < 1) return engine_fail(MI_UNET_EARG, "mi_unet_group_create: a device list needs n_devices >= 1");
        devs.assign(devices, devices + n_devices);
    } else {
        const int n = n_devices <= 0 ? visible - cfg->device : n_devices;
        for (int i = 0; i < n; ++i) devs.push_back(cfg->device + i);
    }
    if (devs.empty()) return engine_fail(MI_UNET_EARG, "mi_unet_group_create: empty device list");
    for (int d : devs)
        if (d < 0 || d >= visible)
            return engine_fail(MI_UNET_EARG, "mi_unet_group_create: device ordinal " + std::to_string(d) + " out of range (" +
                                                 std::to_string(visible) + " visible)");
    auto *g = new mi_unet_group();
    g->cfg = *cfg;
    g->devices = devs;
    for (size_t r = 0; r < devs.size(); ++r) {
        mi_unet_config c = *cfg;
        c.device = devs[r];
        mi_unet_t *h = nullptr;
        if (int rc = mi_unet_create(&c, &h)) { mi_unet_group_destroy(g); return rc; }
        g->eng.push_back(h);
        g->workers.emplace_back(new Worker());
    }
    const size_t R = devs.size();
    g->d_in.assign(R, nullptr); g->d_out.assign(R, nullptr); g->cap_in.assign(R, 0); g->cap_out.assign(R, 0);
    // RCCL communicators: only for > 1 rank on pairwise distinct devices (MIUNET_GROUP_RCCL=0 forces the peer-copy path,
    // =1 also builds a one-rank communicator so that the RCCL calls can be rehearsed on a single GPU)
    const char *env = getenv("MIUNET_GROUP_RCCL");
    // (=2, tests only: communicators even when ranks share a device -- real RCCL refuses that, the stand-in of MIUNET_RCCL_LIB does not)
    const bool distinct = std::set<int>(devs.begin(), devs.end()).size() == R || (env && env[0] == '2');
    const bool want = distinct && !(env && env[0] == '0') && (R > 1 || (env && (env[0] == '1' || env[0] == '2')));
    if (want) {
        std::string why;
        if (g->rccl.load(why)) {
            g->comms.assign(R, nullptr);
            const int rc = g->rccl.CommInitAll(g->comms.data(), (int)R, devs.data());
            if (rc != 0) {
                g->comms.clear();
                if (env && (env[0] == '1' || env[0] == '2')) { const int e = nccl_fail(g, rc, "ncclCommInitAll"); mi_unet_group_destroy(g); return e; }
            } else {
                g->transport = "rccl";
            }
        } else if (env && (env[0] == '1' || env[0] == '2')) {
            mi_unet_group_destroy(g);
            return engine_fail(MI_UNET_EHIP, why);
        }
    }
    *out = g;
    return MI_UNET_OK;
}

int mi_unet_group_clone(mi_unet_group_t *src, mi_unet_group_t **out)
{
    if (!src || !out) return engine_fail(MI_UNET_EARG, "mi_unet_group_clone: null argument");
    *out = nullptr;
    auto *g = new mi_unet_group();
    g->cfg = src->cfg;
    g->devices = src->devices;
    g->transport = src->transport;
    const size_t R = src->eng.size();
    for (size_t r = 0; r < R; ++r) {
        mi_unet_t *h = nullptr;
        if (int rc = mi_unet_clone(src->eng[r], 0, &h)) { mi_unet_group_destroy(g); return rc; }
        g->eng.push_back(h);
        g->workers.emplace_back(new Worker());
    }
    g->d_in.assign(R, nullptr); g->d_out.assign(R, nullptr); g->cap_in.assign(R, 0); g->cap_out.assign(R, 0);
    if (src->postprocess) (void)mi_unet_group_set_postprocess(g, 1);
    *out = g;
    return MI_UNET_OK;
}

int mi_unet_group_size(const mi_unet_group_t *g) { return g ? (int)g->eng.size() : 0; }

mi_unet_t *mi_unet_group_handle(mi_unet_group_t *g, int rank)
{
    return (g && rank >= 0 && rank < (int)g->eng.size()) ? g->eng[rank] : nullptr;
}

const char *mi_unet_group_weight_transport(const mi_unet_group_t *g) { return g ? g->transport.c_str() : ""; }
int mi_unet_group_gather(const mi_unet_group_t *g) { return g ? g->gather : -1; }

int mi_unet_group_set_gather(mi_unet_group_t *g, int mode)
{
    if (!g) return engine_fail(MI_UNET_EARG, "null group");
    if (mode != MI_UNET_GATHER_HOST && mode != MI_UNET_GATHER_XGMI) return engine_fail(MI_UNET_EARG, "unknown gather mode");
    if (mode == MI_UNET_GATHER_XGMI && g->comms.empty())
        return engine_fail(MI_UNET_EARG, "MI_UNET_GATHER_XGMI needs RCCL communicators (distinct devices, librccl loadable)");
    std::lock_guard<std::mutex> lk(g->call_mutex);       // never while a group call is in flight
    g->gather = mode;
    return MI_UNET_OK;
}

int mi_unet_group_set_postprocess(mi_unet_group_t *g, int on)
{
    if (!g) return engine_fail(MI_UNET_EARG, "null group");
    std::lock_guard<std::mutex> lk(g->call_mutex);       // never while a group call is in flight
    g->postprocess = on != 0;
    for (mi_unet_t *h : g->eng)
        if (int rc = mi_unet_set_postprocess(h, on)) return rc;
    return MI_UNET_OK;
}

int mi_unet_group_load_weights_from_memory(mi_unet_group_t *g, const void *blob, size_t len)
{
    if (!g || !blob) return engine_fail(MI_UNET_EARG, "mi_unet_group_load_weights_from_memory: null argument");
    std::lock_guard<std::mutex> lk(g->call_mutex);
    // parse + fold + pack ONCE (seconds of host work for the 31 M-parameter network), upload to the first rank only
    HostWeights hw;
    if (int rc = engine_pack_weights(engine_config(g->eng[0]), engine_algo(g->eng[0]), blob, len, hw)) return rc;
    const int R = (int)g->eng.size();
    for (int r = 0; r < R; ++r)
        if (int rc = engine_adopt_weights(g->eng[r], hw, /*upload=*/r == 0)) return rc;
    if (R == 1 && g->comms.empty()) return engine_calibrate(g->eng[0]);
    const size_t bytes = sizeof(float) * hw.blob.size();
    // Transport ladder for ranks > 0: RCCL broadcast over xGMI -> device-to-device peer copies -> one upload per device from
    // the host blob.  A rung that fails is reported in the transport string and the next one is taken: the weights always
    // arrive, and a node whose RCCL or peer access is misconfigured still runs.
    bool done = false;
    if (const char *v = getenv("MIUNET_GROUP_VERBOSE"); v && v[0] == '1')
        fprintf(stderr, "[miunet group] %d ranks, weights %zu bytes to ranks > 0 by %s\n", R, bytes,
                g->comms.empty() ? "peer copy" : "ncclBroadcast (RCCL)");
    if (!g->comms.empty()) {
        g->transport = "rccl (broadcast in flight)";     // visible to a watchdog if the collective never returns
        // one contiguous broadcast, root = rank 0, every rank on its own engine stream (single-thread group call)
        int rc = g->rccl.GroupStart();
        for (int r = 0; r < R && rc == 0; ++r) {
            if (hipSetDevice(g->devices[r]) != hipSuccess) { rc = -1; break; }
            rc = g->rccl.Broadcast(engine_weight_ptr(g->eng[r]), engine_weight_ptr(g->eng[r]), bytes, Rccl::kUint8, 0, g->comms[r],
                                   engine_stream(g->eng[r]));
        }
        const int rc2 = g->rccl.GroupEnd();
        bool ok = rc == 0 && rc2 == 0;
        for (int r = 0; r < R && ok; ++r) ok = mi_unet_sync(g->eng[r]) == MI_UNET_OK;
        if (ok) { g->transport = "rccl"; done = true; }
        else g->transport = "peer-copy (ncclBroadcast failed)";
    }
    if (!done) {
        bool ok = true;
        for (int r = 1; r < R && ok; ++r)
            ok = hipSetDevice(g->devices[r]) == hipSuccess &&
                 hipMemcpyPeer(engine_weight_ptr(g->eng[r]), g->devices[r], engine_weight_ptr(g->eng[0]), g->devices[0], bytes) == hipSuccess;
        if (ok) done = true;
        else { (void)hipGetLastError(); g->transport = "host-upload (peer copy failed)"; }
    }
    if (!done) {
        for (int r = 1; r < R; ++r) {
            HIP_TRY_G(hipSetDevice(g->devices[r]));
            HIP_TRY_G(hipMemcpy(engine_weight_ptr(g->eng[r]), hw.blob.data(), bytes, hipMemcpyHostToDevice));
        }
    }
    for (int r = 0; r < R; ++r)                          // every rank probes its own copy: same weights, same decision
        if (int rc = engine_calibrate(g->eng[r])) return rc;
    return MI_UNET_OK;
}

int mi_unet_group_load_weights(mi_unet_group_t *g, const char *path)
{
    if (!g || !path) return engine_fail(MI_UNET_EARG, "mi_unet_group_load_weights: null argument");
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.good()) return engine_fail(MI_UNET_EFILE, std::string("Engine file not found: ") + path);
    const std::streamsize sz = f.tellg();
    f.seekg(0);
    std::vector<char> buf((size_t)sz);
    if (!f.read(buf.data(), sz)) return engine_fail(MI_UNET_EFILE, std::string("cannot read ") + path);
    return mi_unet_group_load_weights_from_memory(g, buf.data(), buf.size());
}

int mi_unet_group_infer_u8(mi_unet_group_t *g, const uint8_t *imgs, int B, uint8_t *labels, float *logits)
{
    if (!g || !imgs || !labels || B < 0) return engine_fail(MI_UNET_EARG, "mi_unet_group_infer_u8: bad argument");
    std::lock_guard<std::mutex> lk(g->call_mutex);
    const int R = (int)g->eng.size();
    const size_t hw = (size_t)g->cfg.height * g->cfg.width, in_px = hw * g->cfg.in_ch, lg_px = hw * g->cfg.classes;
    if (g->gather == MI_UNET_GATHER_HOST || logits != nullptr) {
        // every rank: pinned staging -> H2D -> forward -> D2H straight into its range of the caller's buffers
        return for_all_ranks(g, [&](int r) {
            int lo, hi;
            shard_range(B, r, R, lo, hi);
            if (hi == lo) return 0;
            return mi_unet_infer_u8(g->eng[r], imgs + lo * in_px, hi - lo, labels + lo * hw, logits ? logits + lo * lg_px : nullptr);
        });
    }
    // XGMI gather: shards stay on their devices, label maps travel device-to-device into rank 0, one D2H from there
    if (B == 0) return MI_UNET_OK;
    // a failure on one rank must not leave the other ranks' forward passes in flight on buffers the caller may free
    auto drain = [&](int rc_keep) {
        const std::string msg = mi_unet_last_error();
        for (int r = 0; r < R; ++r) (void)mi_unet_sync(g->eng[r]);
        return engine_fail(rc_keep, msg);
    };
    if (int rc = grow(g->d_out[0], g->cap_out[0], (size_t)B * hw, g->devices[0])) return rc;
    if ((size_t)B * hw > g->cap_all) {
        HIP_TRY_G(hipSetDevice(g->devices[0]));
        if (g->h_all) HIP_TRY_G(hipHostFree(g->h_all));
        g->h_all = nullptr; g->cap_all = 0;
        HIP_TRY_G(hipHostMalloc(&g->h_all, (size_t)B * hw, hipHostMallocDefault));
        g->cap_all = (size_t)B * hw;
    }
    int rc = for_all_ranks(g, [&](int r) {
        int lo, hi;
        shard_range(B, r, R, lo, hi);
        const size_t n = (size_t)(hi - lo);
        if (int e = grow(g->d_in[r], g->cap_in[r], n * in_px + 1, g->devices[r])) return e;
        if (r != 0) { if (int e = grow(g->d_out[r], g->cap_out[r], n * hw + 1, g->devices[r])) return e; }
        if (n == 0) return 0;
        HIP_TRY_G(hipSetDevice(g->devices[r]));
        HIP_TRY_G(hipMemcpyAsync(g->d_in[r], imgs + lo * in_px, n * in_px, hipMemcpyHostToDevice, engine_stream(g->eng[r])));
        uint8_t *dst = r == 0 ? g->d_out[0] + lo * hw : g->d_out[r];
        return mi_unet_infer_u8_device(g->eng[r], g->d_in[r], (int)n, dst, nullptr);
    });
    if (rc) return drain(rc);
    rc = g->rccl.GroupStart();
    for (int r = 1; r < R && rc == 0; ++r) {
        int lo, hi;
        shard_range(B, r, R, lo, hi);
        if (hi == lo) continue;
        rc = g->rccl.Send(g->d_out[r], (size_t)(hi - lo) * hw, Rccl::kUint8, 0, g->comms[r], engine_stream(g->eng[r]));
        if (rc == 0) rc = g->rccl.Recv(g->d_out[0] + lo * hw, (size_t)(hi - lo) * hw, Rccl::kUint8, r, g->comms[0], engine_stream(g->eng[0]));
    }
    const int rc2 = g->rccl.GroupEnd();
    if (rc || rc2) return drain(nccl_fail(g, rc ? rc : rc2, "ncclSend/ncclRecv(label maps)"));
    HIP_TRY_G(hipSetDevice(g->devices[0]));
    HIP_TRY_G(hipMemcpyAsync(g->h_all, g->d_out[0], (size_t)B * hw, hipMemcpyDeviceToHost, engine_stream(g->eng[0])));
    for (int r = 0; r < R; ++r)
        if (int e = mi_unet_sync(g->eng[r])) return e;
    memcpy(labels, g->h_all, (size_t)B * hw);
    return MI_UNET_OK;
}

int mi_unet_group_infer_raw16(mi_unet_group_t *g, const uint16_t *const *raws, const int *widths, const int *heights, int B,
                              uint8_t *tiles, uint8_t *labels, float *logits)
{
    if (!g || !raws || !widths || !heights || !labels || B < 0) return engine_fail(MI_UNET_EARG, "mi_unet_group_infer_raw16: bad argument");
    std::lock_guard<std::mutex> lk(g->call_mutex);
    const int R = (int)g->eng.size(), C = g->cfg.in_ch;
    const size_t hw = (size_t)g->cfg.height * g->cfg.width;
    return for_all_ranks(g, [&](int r) {
        int lo, hi;
        shard_range(B, r, R, lo, hi);
        if (hi == lo) return 0;
        return mi_unet_infer_raw16(g->eng[r], raws + (size_t)lo * C, widths + (size_t)lo * C, heights + (size_t)lo * C, hi - lo,
                                   tiles ? tiles + lo * hw * C : nullptr, labels + lo * hw,
                                   logits ? logits + lo * hw * g->cfg.classes : nullptr);
    });
}

int mi_unet_group_segment_raw16(mi_unet_group_t *g, const uint16_t *const *raws, const int *widths, const int *heights, int B,
                                uint8_t *tiles, uint8_t *masks, int32_t *xy, int cap_points, int32_t *start, int cap_contours,
                                int32_t *counts)
{
    if (!g || !raws || !widths || !heights || !masks || !xy || !start || !counts || B < 0 || cap_points <= 0 || cap_contours <= 0)
        return engine_fail(MI_UNET_EARG, "mi_unet_group_segment_raw16: bad argument");
    std::lock_guard<std::mutex> lk(g->call_mutex);
    const int R = (int)g->eng.size(), C = g->cfg.in_ch;
    const size_t hw = (size_t)g->cfg.height * g->cfg.width;
    return for_all_ranks(g, [&](int r) {
        int lo, hi;
        shard_range(B, r, R, lo, hi);
        if (hi == lo) return 0;
        return mi_unet_segment_raw16(g->eng[r], raws + (size_t)lo * C, widths + (size_t)lo * C, heights + (size_t)lo * C, hi - lo,
                                     tiles ? tiles + lo * hw * C : nullptr, masks + lo * hw, xy + (size_t)lo * cap_points * 2, cap_points,
                                     start + (size_t)lo * (cap_contours + 1), cap_contours, counts + lo);
    });
}

void mi_unet_group_destroy(mi_unet_group_t *g)
{
    if (!g) return;
    g->workers.clear();                                  // joins the threads
    for (auto c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    for (size_t r = 0; r < g->d_in.size(); ++r) {
        (void)hipSetDevice(g->devices[r]);
        if (g->d_in[r]) (void)hipFree(g->d_in[r]);
        if (g->d_out[r]) (void)hipFree(g->d_out[r]);
    }
    if (g->h_all) (void)hipHostFree(g->h_all);
    for (mi_unet_t *h : g->eng) mi_unet_destroy(h);
    if (g->rccl.so) dlclose(g->rccl.so);
    delete g;
}

}  // extern "C"
