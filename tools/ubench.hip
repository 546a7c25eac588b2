// Instruction-rate microbenchmarks behind the numbers quoted in DESIGN.md (build: hipcc
// --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/_ubench).  Each kernel runs ITER
// iterations of UNROLL independent dependency chains per lane on every SIMD (8 waves per
// SIMD), so the time per wave-instruction is the pipe's issue interval.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int ITER = 4096;

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed, unsigned long long* stamps)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    // in-kernel clock: shader cycles (s_memtime) against the 100 MHz reference (s_memrealtime)
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    struct Stamp {
        unsigned long long c0, r0, *dst;
        bool on;
        __device__ ~Stamp()
        {
            if (!on) return;
            dst[0] = __builtin_amdgcn_s_memtime() - c0;
            dst[1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
    } stamp{c0, r0, stamps + 2 * blockIdx.x, threadIdx.x == 0};
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
    } else if constexpr (MODE == 7) {  // v_pk_fma_f32: two float32 FMAs per lane and instruction
        f2 a[8];
        for (int i = 0; i < 8; ++i) a[i] = f2{seed + i + tid, seed - i};
        const f2 m{1.000001f, 0.999999f}, c{0.5f, 0.25f};
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
        out[tid] = s;
    } else if constexpr (MODE == 8) {  // v_cndmask_b32 pair (64-bit select) + v_cmp
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed + i + tid;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = (a[i] > 3.0) ? a[i] * 0.5 : 7.0;
            }
        double s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[tid] = (float)s;
    } else if constexpr (MODE == 9) {  // v_sqrt_f32 / v_log_f32 / v_sin_f32 / v_cos_f32 mix (Box-Muller's four)
        float a[8];
        for (int i = 0; i < 8; ++i) a[i] = 0.3f + 1e-3f * i;
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                a[i] = __builtin_amdgcn_sqrtf(a[i]) + 0.1f;
                a[i + 1] = __builtin_amdgcn_logf(a[i + 1]) + 2.0f;
                a[i + 2] = __builtin_amdgcn_sinf(a[i + 2]) + 0.5f;
                a[i + 3] = __builtin_amdgcn_cosf(a[i + 3]) + 0.5f;
            }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[tid] = s;
    }
}

template <int MODE>
static void run(const char* name, double ops_per_iter)
{
    float* out;
    unsigned long long* stamps;
    hipMalloc(&out, sizeof(float) * 256 * 2048);
    hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(2048), dim3(256), 0, 0, out, 1.0f, stamps);
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep)  // ~100 ms of back-to-back launches: the clock has settled
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(2048), dim3(256), 0, 0, out, 1.0f, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20.0f;
    static unsigned long long h[2 * 2048];
    hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int b = 0; b < 2048; ++b) ghz += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
    ghz /= 2048;
    // 2048 blocks x 4 waves = 8192 waves over 1024 SIMDs = 8 waves per SIMD
    const double wave_instr_per_simd = 8.0 * ITER * ops_per_iter;
    printf("%-28s %8.3f ms   %6.2f ns per wave-instruction per SIMD  (= %.1f cycles at the in-kernel clock of %.2f GHz)\n",
           name, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * ghz, ghz);
    hipFree(out);
    hipFree(stamps);
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
    run<7>("v_pk_fma_f32", 8);
    run<8>("f64 cmp+mul+2 cndmask", 32);
    run<9>("sqrt/log/sin/cos f32 (+add)", 16);
    return 0;
}
