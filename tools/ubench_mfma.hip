// Do float32 MFMAs and vector-ALU instructions overlap on a gfx950 SIMD -- inside one wave, and between two
// waves of one SIMD?  (The question behind the trainer's "MFMA + everything else = tile time" in DESIGN 8.2.)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/_ubench_mfma
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int ITER = 20000;

// MODE 0: 4 independent v_mfma_f32_32x32x2_f32 per iteration          (256 matrix-pipe cycles)
// MODE 1: NV independent v_fma_f32 per iteration
// MODE 2: both, interleaved in ONE wave
// MODE 3: waves 0..3 of the workgroup run MODE 0, waves 4..7 MODE 1    (two waves per SIMD)
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, float seed)
{
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(i + r);
    float f[NV];
    for (int i = 0; i < NV; ++i) f[i] = seed + (float)i;
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    for (int it = 0; it < ITER; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) f[i] = __builtin_fmaf(f[i], a, b);
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < NV; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
static double run(const char* name, int threads)
{
    float* out;
    hipMalloc(&out, sizeof(float) * 512 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5.0f;
    printf("%-58s %3d thr  %8.3f ms  = %7.1f ns per iteration\n", name, threads, ms, ms * 1e6 / ITER);
    hipFree(out);
    return ms;
}

int main()
{
    run<0, 32>("4 x mfma_f32_32x32x2, one wave per SIMD", 256);
    run<1, 32>("32 x v_fma_f32, one wave per SIMD", 256);
    run<2, 32>("4 mfma + 32 fma interleaved in ONE wave", 256);
    run<1, 64>("64 x v_fma_f32, one wave per SIMD", 256);
    run<2, 64>("4 mfma + 64 fma interleaved in ONE wave", 256);
    run<0, 32>("4 x mfma, two waves per SIMD", 512);
    run<1, 32>("32 x v_fma_f32, two waves per SIMD", 512);
    run<3, 32>("wave A: 4 mfma | wave B (same SIMD): 32 fma", 512);
    run<3, 64>("wave A: 4 mfma | wave B (same SIMD): 64 fma", 512);
    run<2, 32>("4 mfma + 32 fma in each of two waves per SIMD", 512);
    return 0;
}
