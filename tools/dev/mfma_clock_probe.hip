// mfma_clock_probe.hip -- what the fp32 matrix pipe of THIS card sustains: a register-only v_mfma_f32_16x16x4_f32 loop on every
// CU (one wave per SIMD, 8 independent accumulators), timed with hipEvents, with s_memtime stamps for the shader clock it
// held meanwhile.  Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/probe tools/dev/mfma_clock_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void mfma_loop(float *out, unsigned long long *cycles, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 40000;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * cus * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * cus);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(cus);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= cus;
        const double mfmas = (double)iters * 32.0;                       // per wave
        const double flops = mfmas * 2048.0 * 4.0 * cus;                 // 16x16x4 MACs x 2, four waves per CU
        std::printf("rep %d: %.3f ms, %.1f TFLOP/s, %.2f cycles per MFMA (s_memtime), counter clock %.3f GHz over the kernel, "
                    "clockRate %d kHz, CUs %d\n", rep, ms, flops / (ms * 1e-3) / 1e12, mean / mfmas, mean / (ms * 1e6), p.clockRate, cus);
    }
    return 0;
}
