// Write-stream microbenchmark for the path generator (build: hipcc --offload-arch=gfx950 -O3 tools/ubench_write.hip -o
// tools/_ubench_write).  How fast can the [step][path] float32 matrix be WRITTEN on MI355X, by access pattern?
//   linear   every wave stores consecutive 1 KB pieces of the buffer (grid-stride)
//   walk     the generator's pattern: a thread owns 4 consecutive pair columns p and p + P and stores them row after
//            row (rows ld floats apart), no arithmetic in between
//   walk/xcd the same with the blocks of one XCD owning consecutive columns
//   walk+alu the same as walk with ~the generator's vector work per stored value (a dependent fma chain)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int xcd_block(int bx, int G)
{
    const int r = bx & 7, q = G >> 3, rem = G & 7;
    return r * q + (r < rem ? r : rem) + (bx >> 3);
}

__global__ __launch_bounds__(256) void linear_kernel(float* __restrict__ S, int64_t total)
{
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; j < total; j += stride)
        *reinterpret_cast<float4*>(S + j) = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

template <int XCD, int ALU>
__global__ __launch_bounds__(256) void walk_kernel(float* __restrict__ S, int64_t ld, int64_t P, int N)
{
    const int bx = XCD ? xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int64_t p0 = ((int64_t)bx * 256 + threadIdx.x) * 4;
    if (p0 >= P) return;
    float s[4] = {1.0f, 1.0f, 1.0f, 1.0f}, sa[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float* row = S + p0;
    for (int t = 0; t <= N; ++t) {
        if (ALU) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
#pragma unroll
                for (int k = 0; k < ALU; ++k) {
                    s[v] = __builtin_fmaf(s[v], 0.999f, 0.001f);
                    sa[v] = __builtin_fmaf(sa[v], 1.001f, -0.001f);
                }
            }
        }
        *reinterpret_cast<float4*>(row) = make_float4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<float4*>(row + P) = make_float4(sa[0], sa[1], sa[2], sa[3]);
        row += ld;
    }
}

template <typename F>
static float best_of(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv)
{
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int N = 252;
    const int64_t P = M / 2, total = M * (int64_t)(N + 1);
    float* S;
    if (hipMalloc(&S, sizeof(float) * total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double gb = 4.0 * total / 1e9;
    for (int grid : {1024, 2048, 4096, 8192}) {
        const float ms = best_of([&] { hipLaunchKernelGGL(linear_kernel, dim3(grid), dim3(256), 0, 0, S, total); });
        printf("M=%lld linear grid %5d: %.3f ms  %.2f TB/s\n", (long long)M, grid, ms, gb / ms);
    }
    const int g = (int)((P / 4 + 255) / 256);
    float ms;
    ms = best_of([&] { hipLaunchKernelGGL((walk_kernel<0, 0>), dim3(g), dim3(256), 0, 0, S, M, P, N); });
    printf("M=%lld walk            : %.3f ms  %.2f TB/s\n", (long long)M, ms, gb / ms);
    ms = best_of([&] { hipLaunchKernelGGL((walk_kernel<1, 0>), dim3(g), dim3(256), 0, 0, S, M, P, N); });
    printf("M=%lld walk / xcd      : %.3f ms  %.2f TB/s\n", (long long)M, ms, gb / ms);
    ms = best_of([&] { hipLaunchKernelGGL((walk_kernel<0, 12>), dim3(g), dim3(256), 0, 0, S, M, P, N); });
    printf("M=%lld walk + 24 fma   : %.3f ms  %.2f TB/s\n", (long long)M, ms, gb / ms);
    ms = best_of([&] { hipLaunchKernelGGL((walk_kernel<1, 12>), dim3(g), dim3(256), 0, 0, S, M, P, N); });
    printf("M=%lld walk + 24 fma / xcd: %.3f ms  %.2f TB/s\n", (long long)M, ms, gb / ms);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
