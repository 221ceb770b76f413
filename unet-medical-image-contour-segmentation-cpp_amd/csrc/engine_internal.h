// engine_internal.h -- what group.cpp (the multi-device group of include/mi_unet.h) needs from engine.cpp.  Internal to
// libmiunet.so.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/mi_unet.h"

namespace miunet {

// Weights after BN folding and repacking, in device layout, still on the host: one contiguous blob (one upload or one
// broadcast) plus the offsets the launch plan points at.
struct HostWeights {
    std::vector<float> blob;
    struct Off { size_t w, shift, w4; };        // w4: second packing of the same layer (F(4x4) / per-tap kernels), 0 = none
    std::vector<Off> conv;                      // per 3x3 conv in file order (first one = first-layer layout)
    std::vector<Off> convT;
    Off head{};
};

// The device copy of that blob.  Shared (std::shared_ptr) by a handle and its clones; freed with the last of them.
struct DeviceWeights {
    int device = 0;
    float *d = nullptr;
    size_t floats = 0;
    HostWeights layout;                         // offsets only (blob left empty)
    ~DeviceWeights();
};

int engine_fail(int code, const std::string &msg);                 // sets this thread's mi_unet_last_error()
// parse "MIUNETW1", fold BN, repack for `algo` (a resolved MI_UNET_CONV_* value, see engine_algo)
int engine_pack_weights(const mi_unet_config &cfg, int algo, const void *blob, size_t len, HostWeights &hw);
// allocate the device blob of `h` for `hw` (uploading hw.blob when `upload`, else leaving the bytes to the caller: a
// broadcast or a peer copy fills engine_weight_ptr()), then build the launch plan
int engine_adopt_weights(mi_unet_t *h, const HostWeights &hw, bool upload);
// the numeric guard of the default fp32 plan, run once the weight bytes are on the device (engine.cpp)
int engine_calibrate(mi_unet_t *h);
float *engine_weight_ptr(mi_unet_t *h);
size_t engine_weight_floats(const mi_unet_t *h);
int engine_algo(const mi_unet_t *h);
const mi_unet_config &engine_config(const mi_unet_t *h);
hipStream_t engine_stream(const mi_unet_t *h);

}  // namespace miunet
