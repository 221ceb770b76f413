// wino4_asm.cpp -- host side of conv3x3_wino4a_f32, the hand-scheduled persistent form of the two-block Winograd F(4x4,3x3) kernel.
// The kernel is generated assembly (csrc/asm/gen_wino4_asm.py -> build/wino4a_gfx950.s -> code object), embedded in this library as
// a byte blob (build/wino4a_blob.o) and loaded once per device with hipModuleLoadData.  Same weights (a.wpk4) and the same tensors as
// conv3x3_wino4_f32<2> (csrc/conv_wino4.hip), which stays the fallback for every shape outside the contract below.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"

extern "C" const unsigned char miunet_wino4a_hsaco[];
extern "C" const unsigned char miunet_wino4a_hsaco_end[];
extern "C" const unsigned char miunet_wino4b_hsaco[];          // conv3x3_wino4b_f32 (csrc/asm/gen_wino4b_asm.py): 32 tiles x 64 channels per workgroup
extern "C" const unsigned char miunet_wino4b_hsaco_end[];

namespace miunet {

namespace {

// the kernel's argument block (csrc/asm/gen_wino4_asm.py: s[8:39] after two s_load_dwordx16)
struct Wino4aArgs {
    const float *in, *u, *bias;
    float *out, *pool;
    int32_t H, W, pix_in_bytes, nchunks, tiles_x, tiles_y, m_tiles, nwg;
    uint32_t magic_m, magic_x, magic_y;
    uint32_t u_pos_bytes, u_bytes, img_in_bytes, pix_out_bytes, co_off_bytes, img_out_bytes, pix_pool_bytes, img_pool_bytes;
    float relu_lo;
    int32_t grid, flags;
};
static_assert(sizeof(Wino4aArgs) == 128, "the kernel loads 128 bytes of arguments");

struct PerDevice {
    std::atomic<hipFunction_t> fn{ nullptr };
    hipModule_t mod = nullptr;
    hipError_t err = hipSuccess;
    bool tried = false;
};
PerDevice g_dev[2][64];
std::mutex g_load;

hipError_t function_for_current_device(int which, hipFunction_t *fn)
{
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    PerDevice &d = g_dev[which][dev];
    if (hipFunction_t f = d.fn.load(std::memory_order_acquire)) { *fn = f; return hipSuccess; }
    std::lock_guard<std::mutex> lk(g_load);
    if (hipFunction_t f = d.fn.load(std::memory_order_acquire)) { *fn = f; return hipSuccess; }
    if (d.tried) return d.err;
    d.tried = true;
    const void *image = which == 0 ? miunet_wino4a_hsaco : miunet_wino4b_hsaco;
#ifdef MIUNET_EXPERIMENTS                              // lab build only: a code object FILE instead of the embedded one (same-card A/Bs of kernel variants)
    static std::vector<char> file_images[2];
    std::vector<char> &file_image = file_images[which];
    if (const char *path = getenv(which == 0 ? "MIUNET_WINO4A_HSACO" : "MIUNET_WINO4B_HSACO")) {
        if (FILE *f = fopen(path, "rb")) {
            fseek(f, 0, SEEK_END); file_image.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
            if (fread(file_image.data(), 1, file_image.size(), f) == file_image.size()) image = file_image.data();
            fclose(f);
        }
    }
#endif
    d.err = hipModuleLoadData(&d.mod, image);
    hipFunction_t f = nullptr;
    if (d.err == hipSuccess) d.err = hipModuleGetFunction(&f, d.mod, which == 0 ? "conv3x3_wino4a_f32" : "conv3x3_wino4b_f32");
    if (d.err != hipSuccess) return d.err;
    d.fn.store(f, std::memory_order_release);
    *fn = f;
    return hipSuccess;
}

// ceil(2^32 / d); 0 encodes d == 1.  q = mulhi(n, magic) == n / d for every n with n * d < 2^32.
uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }

}  // namespace

// The shapes the assembly kernel takes: whole 16x16 blocks, an even number (>= 4) of 16-channel K chunks, whole 128-channel
// groups, fp32 in and out, no fused head; every byte offset inside one image below 2^31 and the tile decode exact.
bool conv3x3_wino4a_shape_ok(const ConvArgs &a)
{
    if (a.wpk4 == nullptr || a.head_w != nullptr || a.out_lp) return false;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0 || a.H % 16 || a.W % 16) return false;
    if (a.Cin % 32 || a.Cin < 64 || a.ldc % 4 || a.ldc < a.Cin) return false;
    if (a.Cout % 128 || a.CoutPad < a.Cout || a.ldo % 4 || a.co_off % 4) return false;
    if (a.pool_out != nullptr && a.pool_ld % 4) return false;
    const long long lim = 1ll << 31;
    if ((long long)a.H * a.W * a.ldc * 4 >= lim || (long long)a.H * a.W * a.ldo * 4 >= lim) return false;
    if ((long long)(a.Cin / 16) * 36 * a.CoutPad * 64 >= lim) return false;
    const long long m_tiles = (long long)(a.W / 16) * (a.H / 16) * a.B, nwg = m_tiles * (a.Cout / 128);
    if (nwg >= (1ll << 24) || nwg * m_tiles >= (1ll << 32)) return false;
    return true;
}

hipError_t launch_conv3x3_wino4a(const ConvArgs &a, hipStream_t s)
{
    if (!conv3x3_wino4a_shape_ok(a)) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    if (hipError_t e = function_for_current_device(0, &fn); e != hipSuccess) return e;
    Wino4aArgs k;
    memset(&k, 0, sizeof k);
    k.in = a.in; k.u = a.wpk4; k.bias = a.bias; k.out = a.out; k.pool = a.pool_out;
    k.H = a.H; k.W = a.W; k.pix_in_bytes = a.ldc * 4; k.nchunks = a.Cin / 16;
    k.tiles_x = a.W / 16; k.tiles_y = a.H / 16; k.m_tiles = k.tiles_x * k.tiles_y * a.B; k.nwg = k.m_tiles * (a.Cout / 128);
    k.magic_m = magic_of((uint32_t)k.m_tiles); k.magic_x = magic_of((uint32_t)k.tiles_x); k.magic_y = magic_of((uint32_t)k.tiles_y);
    k.u_pos_bytes = (uint32_t)a.CoutPad * 64u;
    k.u_bytes = (uint32_t)k.nchunks * 36u * k.u_pos_bytes;
    k.img_in_bytes = (uint32_t)a.H * a.W * a.ldc * 4u;
    k.pix_out_bytes = (uint32_t)a.ldo * 4u; k.co_off_bytes = (uint32_t)a.co_off * 4u;
    k.img_out_bytes = (uint32_t)a.H * a.W * a.ldo * 4u;
    const bool pool = a.pool_out != nullptr;
    k.pix_pool_bytes = pool ? (uint32_t)a.pool_ld * 4u : 0u;
    k.img_pool_bytes = pool ? (uint32_t)(a.H / 2) * (a.W / 2) * a.pool_ld * 4u : 0u;
    k.relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    const int cus = routing_of(a).cus;
    k.grid = k.nwg < cus ? k.nwg : cus;             // persistent: one workgroup per CU walks its XCD's tiles
    k.flags = pool ? 1 : 0;
    size_t size = sizeof k;
    void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
    return hipModuleLaunchKernel(fn, (unsigned)k.grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra);
}

// ---- conv3x3_wino4b_f32: blocks of 16 x 32 pixels x 64 channels (the layers with 64 output channels per workgroup)
bool conv3x3_wino4b_shape_ok(const ConvArgs &a)
{
    if (a.wpk4 == nullptr || a.head_w != nullptr || a.out_lp || a.first_img != nullptr) return false;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0 || a.H % 16 || a.W % 32) return false;
    if (a.Cin % 32 || a.Cin < 64 || a.ldc % 4 || a.ldc < a.Cin) return false;
    if (a.Cout % 64 || a.CoutPad < a.Cout || a.ldo % 4 || a.co_off % 4) return false;
    if (a.pool_out != nullptr && a.pool_ld % 4) return false;
    const long long lim = 1ll << 31;
    if ((long long)a.H * a.W * a.ldc * 4 >= lim || (long long)a.H * a.W * a.ldo * 4 >= lim) return false;
    if ((long long)(a.Cin / 16) * 36 * a.CoutPad * 64 >= lim) return false;
    const long long m_tiles = (long long)(a.W / 32) * (a.H / 16) * a.B, nwg = m_tiles * (a.Cout / 64);
    if (nwg >= (1ll << 24) || nwg * m_tiles >= (1ll << 32)) return false;
    return true;
}

hipError_t launch_conv3x3_wino4b(const ConvArgs &a, hipStream_t s)
{
    if (!conv3x3_wino4b_shape_ok(a)) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    if (hipError_t e = function_for_current_device(1, &fn); e != hipSuccess) return e;
    Wino4aArgs k;
    memset(&k, 0, sizeof k);
    k.in = a.in; k.u = a.wpk4; k.bias = a.bias; k.out = a.out; k.pool = a.pool_out;
    k.H = a.H; k.W = a.W; k.pix_in_bytes = a.ldc * 4; k.nchunks = a.Cin / 16;
    k.tiles_x = a.W / 32; k.tiles_y = a.H / 16; k.m_tiles = k.tiles_x * k.tiles_y * a.B; k.nwg = k.m_tiles * (a.Cout / 64);
    k.magic_m = magic_of((uint32_t)k.m_tiles); k.magic_x = magic_of((uint32_t)k.tiles_x); k.magic_y = magic_of((uint32_t)k.tiles_y);
    k.u_pos_bytes = (uint32_t)a.CoutPad * 64u;
    k.u_bytes = (uint32_t)k.nchunks * 36u * k.u_pos_bytes;
    k.img_in_bytes = (uint32_t)a.H * a.W * a.ldc * 4u;
    k.pix_out_bytes = (uint32_t)a.ldo * 4u; k.co_off_bytes = (uint32_t)a.co_off * 4u;
    k.img_out_bytes = (uint32_t)a.H * a.W * a.ldo * 4u;
    const bool pool = a.pool_out != nullptr;
    k.pix_pool_bytes = pool ? (uint32_t)a.pool_ld * 4u : 0u;
    k.img_pool_bytes = pool ? (uint32_t)(a.H / 2) * (a.W / 2) * a.pool_ld * 4u : 0u;
    k.relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    const int cus = routing_of(a).cus;
    k.grid = k.nwg < cus ? k.nwg : cus;
    k.flags = pool ? 1 : 0;
    size_t size = sizeof k;
    void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
    return hipModuleLaunchKernel(fn, (unsigned)k.grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra);
}

}  // namespace miunet
