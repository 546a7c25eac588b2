// Instruction-rate microbenchmarks behind the numbers quoted in DESIGN.md (build: hipcc
// --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/_ubench).  Each kernel runs ITER
// iterations of UNROLL independent dependency chains per lane on every SIMD (8 waves per
// SIMD), so the time per wave-instruction is the pipe's issue interval.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if constexpr (MODE == 0) {  // v_fma_f32
        float a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.000001f, 0.5f);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[tid] = s;
    } else if constexpr (MODE == 1) {  // v_fma_f64
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], 1.000001, 0.5);
        double s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[tid] = (float)s;
    } else if constexpr (MODE == 2) {  // v_add_f64
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] + 0.5;
        double s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[tid] = (float)s;
    } else if constexpr (MODE == 3) {  // v_mad_u64_u32 (what the Philox round compiles to)
        uint32_t a[8];
        for (int i = 0; i < 8; ++i) a[i] = (uint32_t)seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint64_t p = (uint64_t)0xD2511F53u * a[i];
                a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p;
            }
        uint32_t s = 0;
        for (int i = 0; i < 8; ++i) s ^= a[i];
        out[tid] = (float)s;
    } else if constexpr (MODE == 4) {  // v_mul_hi_u32 + v_mul_lo_u32 forced apart
        uint32_t a[8];
        for (int i = 0; i < 8; ++i) a[i] = (uint32_t)seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t hi = __umulhi(0xD2511F53u, a[i]);
                uint32_t lo;
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(0xD2511F53u));
                a[i] = hi ^ lo;
            }
        uint32_t s = 0;
        for (int i = 0; i < 8; ++i) s ^= a[i];
        out[tid] = (float)s;
    } else if constexpr (MODE == 5) {  // v_exp_f32
        float a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed + i * 1e-3f;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.25f;
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[tid] = s;
    } else if constexpr (MODE == 6) {  // v_cvt_f64_f32
        float a[8];
        double acc = 0;
        for (int i = 0; i < 8; ++i) a[i] = seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double d = (double)a[i];
                asm volatile("" : "+v"(d));
                a[i] = (float)d + 1.0f;
            }
        for (int i = 0; i < 8; ++i) acc += a[i];
        out[tid] = (float)acc;
    }
}

template <int MODE>
static void run(const char* name, double ops_per_iter)
{
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(2048), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(2048), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 2048 blocks x 4 waves = 8192 waves over 1024 SIMDs = 8 waves per SIMD
    const double wave_instr_per_simd = 8.0 * ITER * ops_per_iter;
    printf("%-28s %8.3f ms   %6.2f ns per wave-instruction per SIMD  (= %.1f cycles at 2.4 GHz)\n", name,
           ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32", 8);
    run<1>("v_fma_f64", 8);
    run<2>("v_add_f64", 8);
    run<3>("v_mad_u64_u32 (+xor)", 8);
    run<4>("v_mul_hi_u32+v_mul_lo_u32", 16);
    run<5>("v_exp_f32 (+mul)", 8);
    run<6>("v_cvt_f64_f32+cvt_f32_f64", 16);
    return 0;
}
