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
    unsigned *d; hipMalloc(&d, 2 * 256 * 64); hipMemset(d, 0xEE, 2 * 256 * 64);
    struct { unsigned *p; } k{ d };
    size_t size = sizeof k;
    void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
    if (hipModuleLaunchKernel(fn, 2, 1, 1, 256, 1, 1, 0, nullptr, nullptr, extra) != hipSuccess) { printf("launch failed\n"); return 2; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 2; }
    std::vector<unsigned> h(2 * 256 * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < 512; t += 64) {
        printf("slot %3d readfirstlane:", t); for (int i = 0; i < 10; ++i) printf(" %08x", h[t * 16 + i]);
        printf("\n         s_add      :"); for (int i = 0; i < 10; ++i) printf(" %08x", h[t * 16 + 10 + i]);
        printf("\n");
    }
    return 0;
}
