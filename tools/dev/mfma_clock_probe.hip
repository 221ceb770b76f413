// mfma_clock_probe.hip -- what the matrix pipes of THIS card sustain: register-only MFMA loops on every CU (one wave per SIMD,
// independent accumulators), timed with hipEvents, with s_memtime stamps for the shader clock held meanwhile.
//   fp32 : v_mfma_f32_16x16x4_f32   (the instruction of the Winograd kernels)
//   bf16 : v_mfma_f32_32x32x16_bf16 (the instruction of the 16-bit kernels until round 3) and v_mfma_f32_16x16x32_bf16 (since): the same
//          FLOPs per cycle; which one the chip clocks higher on random operands is the question (MI355X_MICROARCH.md, DVFS give-back 7)
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/probe tools/dev/mfma_clock_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 1) void mfma_f32_loop(float *out, unsigned long long *cycles, int iters)
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

__global__ __launch_bounds__(256, 1) void mfma_bf16_loop(float *out, unsigned long long *cycles, int iters, int seed)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) {                       // non-trivial operands: the data path toggles as it does on real tensors
        a[k] = (__bf16)(0.01f * (float)((threadIdx.x * 7 + k * 13 + seed) % 97) - 0.4f);
        b[k] = (__bf16)(0.02f * (float)((threadIdx.x * 5 + k * 11 + seed) % 89) - 0.7f);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// the same 64 accumulator registers as 16 blocks of 16 x 16, accumulated in place (csrc/lpr_common.h)
__global__ __launch_bounds__(256, 1) void mfma_bf16_loop16(float *out, unsigned long long *cycles, int iters, int seed)
{
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) {
        a[k] = (__bf16)(0.01f * (float)((threadIdx.x * 7 + k * 13 + seed) % 97) - 0.4f);
        b[k] = (__bf16)(0.02f * (float)((threadIdx.x * 5 + k * 11 + seed) % 89) - 0.7f);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * cus * 256);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * cus);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind) {
        const int iters = kind == 0 ? 40000 : 60000;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(mfma_f32_loop, dim3(cus), dim3(256), 0, 0, out, cyc, iters);
            else if (kind == 1) hipLaunchKernelGGL(mfma_bf16_loop, dim3(cus), dim3(256), 0, 0, out, cyc, iters, rep);
            else hipLaunchKernelGGL(mfma_bf16_loop16, dim3(cus), dim3(256), 0, 0, out, cyc, iters, rep);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(cus);
            (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= cus;
            const double mfmas = (double)iters * (kind == 2 ? 64.0 : 32.0);             // per wave
            const double flop_per = kind == 0 ? 2048.0 : kind == 1 ? 32768.0 : 16384.0;   // 16x16x4 / 32x32x16 / 16x16x32 MACs x 2
            std::printf("%s rep %d: %.3f ms, %.1f TFLOP/s, %.2f cycles per MFMA (s_memtime), counter clock %.3f GHz over the kernel\n",
                        kind == 0 ? "fp32 16x16x4 " : kind == 1 ? "bf16 32x32x16" : "bf16 16x16x32", rep, ms, mfmas * flop_per * 4.0 * cus / (ms * 1e-3) / 1e12,
                        mean / mfmas, mean / (ms * 1e6));
        }
    }
    return 0;
}
