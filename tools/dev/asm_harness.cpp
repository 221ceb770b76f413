// asm_harness.cpp -- bring-up of the assembly F(4x4,3x3) kernel outside the library: loads a code object FILE, builds the argument
// block the way csrc/wino4_asm.cpp does, launches once on random data and reports whether the launch completed and how much of the
// NaN-poisoned output was written.  One process per run: a fault ends this process only.
//   hipcc -O2 -o /tmp/asm_harness tools/dev/asm_harness.cpp && /tmp/asm_harness file.hsaco B H W Cin Cout [pool]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Args {
    const float *in, *u, *bias;
    float *out, *pool;
    int32_t H, W, pix_in_bytes, nchunks, tiles_x, tiles_y, m_tiles, nwg;
    uint32_t magic_m, magic_x, magic_y;
    uint32_t u_pos_bytes, u_bytes, img_in_bytes, pix_out_bytes, co_off_bytes, img_out_bytes, pix_pool_bytes, img_pool_bytes;
    float relu_lo;
    int32_t grid, flags;
};
static uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 7) { std::printf("usage: asm_harness file.hsaco B H W Cin Cout [pool]\n"); return 2; }
    const int B = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), Cin = atoi(argv[5]), Cout = atoi(argv[6]);
    const bool pool = argc > 7 && std::string(argv[7]) == "pool";
    const bool stamps = argc > 7 && std::string(argv[7]) == "stamps";   // a --stamps build: the POOL pointer receives [wg][wave][8] cycle sums
    const char *dump_dir = argc > 8 ? argv[8] : nullptr;          // writes in.bin, u.bin, bias.bin, out.bin there
    std::vector<char> blob;
    if (FILE *f = fopen(argv[1], "rb")) { fseek(f, 0, SEEK_END); blob.resize(ftell(f)); fseek(f, 0, SEEK_SET); if (fread(blob.data(), 1, blob.size(), f) != blob.size()) return 2; fclose(f); }
    else { std::printf("cannot read %s\n", argv[1]); return 2; }
    hipModule_t mod; hipFunction_t fn;
    CK(hipModuleLoadData(&mod, blob.data()));
    const bool mode_b = getenv("ASM_MODE") && getenv("ASM_MODE")[0] == 'b';       // conv3x3_wino4b_f32: blocks of 16 x 32 pixels x 64 channels
    CK(hipModuleGetFunction(&fn, mod, mode_b ? "conv3x3_wino4b_f32" : "conv3x3_wino4a_f32"));
    const size_t in_n = (size_t)B * H * W * Cin, out_n = (size_t)B * H * W * Cout, u_n = (size_t)(Cin / 16) * 36 * Cout * 16;
    std::vector<float> h_in(in_n), h_u(u_n), h_b(Cout);
    uint32_t x = 12345;
    auto rnd = [&] { x = x * 1664525u + 1013904223u; return (float)((int)(x >> 20) - 2048) / 2048.f; };
    for (float &v : h_in) v = rnd();
    for (float &v : h_u) v = rnd() * 0.05f;
    for (float &v : h_b) v = rnd();
    float *d_in, *d_u, *d_b, *d_out, *d_pool = nullptr;
    CK(hipMalloc(&d_in, in_n * 4)); CK(hipMalloc(&d_u, u_n * 4)); CK(hipMalloc(&d_b, Cout * 4)); CK(hipMalloc(&d_out, out_n * 4));
    if (pool) { CK(hipMalloc(&d_pool, out_n)); CK(hipMemset(d_pool, 0xFF, out_n)); }
    if (stamps) { CK(hipMalloc(&d_pool, 1 << 20)); CK(hipMemset(d_pool, 0, 1 << 20)); }
    CK(hipMemcpy(d_in, h_in.data(), in_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_u, h_u.data(), u_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xFF, out_n * 4));
    Args k;
    memset(&k, 0, sizeof k);
    k.in = d_in; k.u = d_u; k.bias = d_b; k.out = d_out; k.pool = d_pool;
    k.H = H; k.W = W; k.pix_in_bytes = Cin * 4; k.nchunks = Cin / 16;
    k.tiles_x = W / (mode_b ? 32 : 16); k.tiles_y = H / 16; k.m_tiles = k.tiles_x * k.tiles_y * B; k.nwg = k.m_tiles * (Cout / (mode_b ? 64 : 128));
    k.magic_m = magic_of(k.m_tiles); k.magic_x = magic_of(k.tiles_x); k.magic_y = magic_of(k.tiles_y);
    k.u_pos_bytes = Cout * 64u; k.u_bytes = k.nchunks * 36u * k.u_pos_bytes;
    k.img_in_bytes = (uint32_t)H * W * Cin * 4u; k.pix_out_bytes = Cout * 4u; k.co_off_bytes = 0; k.img_out_bytes = (uint32_t)H * W * Cout * 4u;
    k.pix_pool_bytes = pool ? Cout * 4u : 0; k.img_pool_bytes = pool ? (uint32_t)(H / 2) * (W / 2) * Cout * 4u : 0;
    k.relu_lo = 0.f;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    k.grid = k.nwg < p.multiProcessorCount ? k.nwg : p.multiProcessorCount;
    if (const char *g = getenv("ASM_GRID")) k.grid = atoi(g) < k.grid ? atoi(g) : k.grid;      // fewer workgroups than CUs: what the chip-wide bursts cost
    k.flags = pool ? 1 : 0;
    size_t size = sizeof k;
    void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    CK(hipModuleLaunchKernel(fn, k.grid, 1, 1, 256, 1, 1, 0, nullptr, nullptr, extra));
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (stamps) {
        for (int rep = 0; rep < 3; ++rep) {          // warm clocks and caches, then the measured launch
            CK(hipMemset(d_pool, 0, 1 << 20));
            CK(hipEventRecord(e0));
            CK(hipModuleLaunchKernel(fn, k.grid, 1, 1, 256, 1, 1, 0, nullptr, nullptr, extra));
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<uint32_t> st((size_t)k.grid * 4 * 8);
        CK(hipMemcpy(st.data(), d_pool, st.size() * 4, hipMemcpyDeviceToHost));
        const char *names[6] = { "first body", "mid bodies", "last body", "first transform (next tile)", "epilogue", "join" };
        const double tiles = (double)k.nwg / k.grid;
        std::printf("stamps, %d tiles over %d workgroups (%.2f per workgroup), K = %d chunks, launch %.3f ms; cycles per TILE, mean over workgroups:\n", k.nwg, k.grid, tiles, k.nchunks, ms);
        for (int w = 0; w < 4; w += 2) {
            double tot = 0;
            std::printf("  wave %d (%s):", w, w == 0 ? "two-row role" : "one-row role");
            for (int ph = 0; ph < 6; ++ph) {
                double sum = 0;
                for (int g = 0; g < k.grid; ++g) sum += st[((size_t)g * 4 + w) * 8 + ph];
                sum /= k.grid * tiles;
                tot += sum;
                std::printf(" %s %.0f%s", names[ph], sum, ph == 1 && k.nchunks > 2 ? (std::string(" (") + std::to_string((int)(sum / (k.nchunks - 2))) + " per chunk)").c_str() : "");
                std::printf(";");
            }
            std::printf(" total %.0f\n", tot);
        }
    }
    std::vector<float> h_out(out_n);
    CK(hipMemcpy(h_out.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
    size_t nan = 0; double sum = 0;
    for (float v : h_out) { if (std::isnan(v)) ++nan; else sum += v; }
    if (dump_dir) {
        auto put = [&](const char *name, const void *p, size_t bytes) {
            const std::string path = std::string(dump_dir) + "/" + name;
            if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(p, 1, bytes, f); fclose(f); }
        };
        put("in.bin", h_in.data(), in_n * 4); put("u.bin", h_u.data(), u_n * 4); put("bias.bin", h_b.data(), Cout * 4); put("out.bin", h_out.data(), out_n * 4);
    }
    std::printf("%s %dx%dx%dx%d->%d grid %d: completed in %.3f ms, unwritten outputs %zu of %zu, checksum %.6g\n", argv[1], B, H, W, Cin, Cout, k.grid, ms, nan, out_n, sum);
    return 0;
}
