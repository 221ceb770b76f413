// fp32_mfma_filler_probe.hip -- what one instruction costs NEXT TO v_mfma_f32_16x16x4_f32 when a wave is alone on its SIMD (the
// regime of conv3x3_wino4_f32<2>: 288 accumulators, one wave per SIMD).  A loop of 32 independent-accumulator MFMAs per iteration
// with N fillers of one kind in every MFMA gap, timed with s_memtime; the bare loop is the 32-cycle floor.  Answers, for the
// hand-scheduled K loop: which instructions hide in an fp32 MFMA's shadow (SQ_VALU_MFMA_COEXEC_CYCLES is 0 on the fp32 kernels),
// what a packed-f32 VALU instruction costs against two plain ones, and what LDS / VMEM / SALU / s_waitcnt issue costs there.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/fprobe tools/dev/fp32_mfma_filler_probe.hip && /tmp/fprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { NONE, V_FMA, V_PK_FMA, V_PK_ADD, V_ADD, V_MOV, DS_READ128, DS_WRITE128, BUF_LOAD128, S_NOP, S_ADD, ACC_WRITE, ACC_READ, S_WAIT_NOOP, V_PK_MUL, LDS_DMA, KINDS };
static const char *NAMES[KINDS] = { "none", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_add_f32", "v_mov_b32", "ds_read_b128", "ds_write_b128",
                                    "buffer_load_dwordx4 (L2 hit)", "s_nop 0", "s_add_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32",
                                    "s_waitcnt (already satisfied)", "v_pk_mul_f32", "buffer_load_dwordx4 lds (L2 hit)" };

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float *out, unsigned long long *cycles, const float *src, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    float x0 = a, x1 = b, x2 = 0.25f, x3 = 0.125f;
    f32x2 p0 = { a, b }, p1 = { b, a }, p2 = { 0.5f, 0.25f };
    f32x4 ld[4] = {};
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    float *lp = lds + threadIdx.x * 4;
    const float *gp = src + threadIdx.x * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 20, 0x00020000);
    unsigned sacc = 0;
    float ar = 0.f;
    const unsigned voff = threadIdx.x * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if constexpr (KIND == V_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(n & 1 ? x2 : x3) : "v"(x0), "v"(x1));
                if constexpr (KIND == V_ADD) asm volatile("v_add_f32 %0, %1, %0" : "+v"(n & 1 ? x2 : x3) : "v"(x0));
                if constexpr (KIND == V_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(n & 1 ? x2 : x3) : "v"(x0));
                if constexpr (KIND == V_PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(n & 1 ? p2 : p1) : "v"(p0), "v"(p0));
                if constexpr (KIND == V_PK_ADD) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(n & 1 ? p2 : p1) : "v"(p0));
                if constexpr (KIND == V_PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(n & 1 ? p2 : p1) : "v"(p0));
                if constexpr (KIND == DS_READ128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(i + n) & 3]) : "v"((unsigned)(size_t)lp) : "memory");
                if constexpr (KIND == DS_WRITE128) asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(size_t)lp), "v"(ld[0]) : "memory");
                if constexpr (KIND == BUF_LOAD128) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[(i + n) & 3]) : "v"(voff), "s"(rsrc) : "memory");
                if constexpr (KIND == LDS_DMA) asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(4096 * ((i + n) & 3)) : "memory");
                if constexpr (KIND == S_NOP) asm volatile("s_nop 0");
                if constexpr (KIND == S_ADD) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                if constexpr (KIND == ACC_WRITE) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ar) : "v"(x0));
                if constexpr (KIND == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(n & 1 ? x2 : x3) : "a"(ar));
                if constexpr (KIND == S_WAIT_NOOP) asm volatile("s_waitcnt vmcnt(63)");
            }
        }
        if constexpr (KIND == DS_READ128 || KIND == DS_WRITE128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == BUF_LOAD128 || KIND == LDS_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = x2 + x3 + p1[0] + p1[1] + p2[0] + p2[1] + ar + (float)sacc;
    for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][1] + ld[i][2] + ld[i][3];
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[(threadIdx.x * 7) & 8191] + gp[0];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int N>
static void run(int cus, float *out, unsigned long long *cyc, const float *src, double floor_cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((probe<KIND, N>), dim3(cus), dim3(256), 0, 0, out, cyc, src, iters);      // warm-up
    hipLaunchKernelGGL((probe<KIND, N>), dim3(cus), dim3(256), 0, 0, out, cyc, src, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cus);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= cus;
    const double per = mean / (iters * 32.0);
    std::printf("%-34s x%d per gap: %7.2f cycles per MFMA  (+%6.2f over the bare loop = %5.2f per filler)\n", NAMES[KIND], N, per,
                per - floor_cyc, N ? (per - floor_cyc) / N : 0.0);
    std::fflush(stdout);
}

template <int KIND>
static void sweep(int cus, float *out, unsigned long long *cyc, const float *src, double floor_cyc)
{
    run<KIND, 1>(cus, out, cyc, src, floor_cyc);
    run<KIND, 2>(cus, out, cyc, src, floor_cyc);
    run<KIND, 4>(cus, out, cyc, src, floor_cyc);
    run<KIND, 6>(cus, out, cyc, src, floor_cyc);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out, *src;
    unsigned long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * cus * 256);
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * cus);
    run<NONE, 0>(cus, out, cyc, src, 32.0);
    const double fl = 32.0;
    sweep<V_FMA>(cus, out, cyc, src, fl);
    sweep<V_ADD>(cus, out, cyc, src, fl);
    sweep<V_MOV>(cus, out, cyc, src, fl);
    sweep<V_PK_FMA>(cus, out, cyc, src, fl);
    sweep<V_PK_ADD>(cus, out, cyc, src, fl);
    sweep<V_PK_MUL>(cus, out, cyc, src, fl);
    sweep<ACC_WRITE>(cus, out, cyc, src, fl);
    sweep<ACC_READ>(cus, out, cyc, src, fl);
    sweep<S_NOP>(cus, out, cyc, src, fl);
    sweep<S_ADD>(cus, out, cyc, src, fl);
    sweep<S_WAIT_NOOP>(cus, out, cyc, src, fl);
    run<DS_READ128, 1>(cus, out, cyc, src, fl);
    run<DS_READ128, 2>(cus, out, cyc, src, fl);
    run<DS_WRITE128, 1>(cus, out, cyc, src, fl);
    run<DS_WRITE128, 2>(cus, out, cyc, src, fl);
    run<BUF_LOAD128, 1>(cus, out, cyc, src, fl);
    run<BUF_LOAD128, 2>(cus, out, cyc, src, fl);
    run<LDS_DMA, 1>(cus, out, cyc, src, fl);
    return 0;
}
