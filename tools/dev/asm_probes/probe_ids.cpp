// loads probe_ids.hsaco, launches 2 workgroups of 256, prints per wave: raw v0, readfirstlane((v0 & 0x3ff) >> 6), exec, lane id, wg id
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
int main(int argc, char **argv)
{
    std::vector<char> blob;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); blob.resize(ftell(f)); fseek(f, 0, SEEK_SET); if (fread(blob.data(), 1, blob.size(), f) != blob.size()) return 2; fclose(f);
    hipModule_t mod; hipFunction_t fn;
    if (hipModuleLoadData(&mod, blob.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "probe_ids") != hipSuccess) { printf("load failed\n"); return 2; }
    unsigned *d; hipMalloc(&d, 2 * 256 * 32); hipMemset(d, 0xEE, 2 * 256 * 32);
    struct { unsigned *p; } k{ d };
    size_t size = sizeof k;
    void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
    if (hipModuleLaunchKernel(fn, 2, 1, 1, 256, 1, 1, 0, nullptr, nullptr, extra) != hipSuccess) { printf("launch failed\n"); return 2; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 2; }
    std::vector<unsigned> h(2 * 256 * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < 512; t += 16)
        printf("slot %3d: v0 %08x  wave(readfirstlane) %u  exec %08x%08x  lane %u  wg %u  v0&3ff %u  >>6 %u\n", t, h[t * 8], h[t * 8 + 1], h[t * 8 + 3], h[t * 8 + 2], h[t * 8 + 4], h[t * 8 + 5], h[t * 8 + 6], h[t * 8 + 7]);
    return 0;
}
