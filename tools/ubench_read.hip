// Read-stream microbenchmark for the LSM sweeps (build: hipcc --offload-arch=gfx950 -O3 tools/ubench_read.hip -o
// tools/_ubench_read).  How fast can 1 GB of the [step][path] float32 matrix be READ on MI355X, by access pattern?
//   linear     every wave walks consecutive 4 KB pieces of the buffer (grid-stride)
//   rows       the pass-1 pattern: a wave owns 1024 consecutive paths and walks down 32 rows (4 MB apart)
//   rows_lds   the same pattern through LDS-DMA (global_load_lds_dwordx4: no VGPR landing zone)
// Loads are nontemporal 16-byte accesses, three pieces in flight per lane group as in lsm_pass1_kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ldnt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)); }

// MODE 0 linear, 1 rows; PAD_KB of LDS per workgroup caps the workgroups per CU (160 KB / PAD_KB)
template <int MODE, int PAD_KB = 0>
__global__ __launch_bounds__(256) void read_kernel(const float* __restrict__ S, int64_t M, int N, int tchunk, float* sink)
{
    __shared__ float pad[PAD_KB > 0 ? PAD_KB * 256 : 1];
    if (PAD_KB > 0 && M < 0) pad[threadIdx.x] = 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t fold = 0;
    auto eat = [&](const f4 (&b)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            fold ^= __float_as_uint(b[k].x) ^ __float_as_uint(b[k].y) ^ __float_as_uint(b[k].z) ^ __float_as_uint(b[k].w);
    };
    if (MODE == 0) {
        // 16 KB per workgroup per step, all workgroups side by side, then the next 16 KB stripe
        const int64_t total = M * (int64_t)(N + 1);
        const int64_t wg_stride = (int64_t)gridDim.x * 4096;
        int64_t j = ((int64_t)blockIdx.x * 4 + wave) * 1024 + lane * 4;
        f4 a[4], b[4], c[4];
        auto ld = [&](f4 (&d)[4], int64_t jj) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = ldnt(S + ((jj + k * 256 < total) ? jj + k * 256 : 0));
        };
        ld(a, j); ld(b, j + wg_stride);
        for (; j < total; j += 3 * wg_stride) {
            ld(c, j + 2 * wg_stride); eat(a);
            ld(a, j + 3 * wg_stride); eat(b);
            ld(b, j + 4 * wg_stride); eat(c);
        }
    } else {
        const int64_t ntiles = (M + 1023) / 1024;
        const int64_t tg = (int64_t)blockIdx.x * 4 + wave;
        if (tg >= ntiles) return;
        const int t0 = 1 + blockIdx.y * tchunk, t1 = min(t0 + tchunk, N);
        const float* col = S + tg * 1024 + lane * 4;
        f4 a[4], b[4], c[4];
        auto ld = [&](f4 (&d)[4], int t) {
            t = min(t, t1 - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = ldnt(col + (int64_t)t * M + k * 256);
        };
        ld(a, t0); ld(b, t0 + 1);
        for (int t = t0; t < t1; t += 3) {
            ld(c, t + 2); eat(a);
            ld(a, t + 3); eat(b);
            ld(b, t + 4); eat(c);
        }
    }
    if (fold == 0x12345678u) sink[threadIdx.x] = 1.0f;
}

// the rows pattern through LDS-DMA: one 1 KB piece per wave-instruction lands in LDS, ds_read picks it up
__global__ __launch_bounds__(256) void read_lds_kernel(const float* __restrict__ S, int64_t M, int N, int tchunk, float* sink)
{
    __shared__ float buf[4][3][4 * 256];  // [wave][ring slot][4 pieces x 256 floats] = 48 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntiles = (M + 1023) / 1024;
    const int64_t tg = (int64_t)blockIdx.x * 4 + wave;
    if (tg >= ntiles) return;
    const int t0 = 1 + blockIdx.y * tchunk, t1 = min(t0 + tchunk, N);
    const float* col = S + tg * 1024 + lane * 4;
    uint32_t fold = 0;
    auto issue = [&](int slot, int t) {
        t = min(t, t1 - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* g = col + (int64_t)t * M + k * 256;
            const uint32_t dst = (uint32_t)(uintptr_t)(&buf[wave][slot][k * 256]);  // wave-uniform LDS byte address
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
        }
    };
    auto eat = [&](int slot) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f4 v = *reinterpret_cast<const f4*>(&buf[wave][slot][k * 256 + lane * 4]);
            fold ^= __float_as_uint(v.x) ^ __float_as_uint(v.y) ^ __float_as_uint(v.z) ^ __float_as_uint(v.w);
        }
    };
    issue(0, t0); issue(1, t0 + 1);
    int slot = 0;
    for (int t = t0; t < t1; ++t) {
        issue((slot + 2) % 3, t + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // all but the two youngest rows (4 pieces each) have landed
        eat(slot);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is read before it is refilled next iteration
        slot = (slot + 1) % 3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (fold == 0x12345678u) sink[threadIdx.x] = 1.0f;
}

// the pass-2 pattern: a thread owns Q x 4 consecutive-in-tile paths for the WHOLE sweep and walks the rows
// N-1 .. 1 in blocks of U (U x Q independent 16-byte loads in flight per lane, nothing prefetched across blocks);
// a wave touches Q KB per row.  NT: nontemporal hint on or off.
template <int Q, int U, bool NT>
__global__ __launch_bounds__(256) void walk_kernel(const float* __restrict__ S, int64_t M, int N, float* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = ((int64_t)blockIdx.x * 4 + wave) * (256 * Q);
    if (tile >= M) return;
    const float* col = S + tile + lane * 4;
    uint32_t fold = 0;
    int t = N - 1;
    for (; t >= U; t -= U) {
        f4 b[U][Q];
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const f4* p = reinterpret_cast<const f4*>(col + (int64_t)(t - k) * M + q * 256);
                b[k][q] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int q = 0; q < Q; ++q)
                fold ^= __float_as_uint(b[k][q].x) ^ __float_as_uint(b[k][q].y) ^ __float_as_uint(b[k][q].z) ^ __float_as_uint(b[k][q].w);
    }
    if (fold == 0x12345678u) sink[threadIdx.x] = 1.0f;
}

template <int Q, int U, bool NT>
static void run_walk(const float* S, int64_t M, int N, float* sink, hipEvent_t e0, hipEvent_t e1)
{
    const unsigned grid = (unsigned)((M + 1024 * Q - 1) / (1024 * Q));
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((walk_kernel<Q, U, NT>), dim3(grid), dim3(256), 0, 0, S, M, N, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4 && ms < best) best = ms;
    }
    const double b = 4.0 * M * (N - 1 - (N - 1) % U);
    printf("walk  %d KB per wave-row, %2d rows in flight, %s, %5u workgroups: %7.3f ms  %6.2f TB/s\n", Q, U,
           NT ? "nt" : "  ", grid, best, b / best / 1e9);
}

int main(int argc, char** argv)
{
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int N = 252, tchunk = 32;
    float *S, *sink;
    hipMalloc(&S, sizeof(float) * M * (N + 1));
    hipMalloc(&sink, 4096);
    hipMemset(S, 0x3f, sizeof(float) * M * (N + 1));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = 4.0 * M * N;
    const dim3 grows((unsigned)((M / 1024 + 1 + 3) / 4), (N - 1 + tchunk - 1) / tchunk);
    for (int occ : {8, 6, 4, 3, 2}) {  // workgroups (= waves per SIMD) per CU, limited through LDS
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            hipEventRecord(e0);
            if (occ == 8) hipLaunchKernelGGL((read_kernel<1, 20>), grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
            else if (occ == 6) hipLaunchKernelGGL((read_kernel<1, 26>), grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
            else if (occ == 4) hipLaunchKernelGGL((read_kernel<1, 40>), grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
            else if (occ == 3) hipLaunchKernelGGL((read_kernel<1, 53>), grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
            else hipLaunchKernelGGL((read_kernel<1, 80>), grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 3 && ms < best) best = ms;
        }
        printf("rows, %d workgroups per CU: %7.3f ms  %6.2f TB/s\n", occ, best, bytes / best / 1e9);
    }
    for (int mode = 0; mode < 3; ++mode) {
        for (int grid0 : {1024, 2048, 4096}) {
            if (mode != 0 && grid0 != 1024) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(read_kernel<0>, dim3(grid0), dim3(256), 0, 0, S, M, N, tchunk, sink);
                else if (mode == 1) hipLaunchKernelGGL(read_kernel<1>, grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
                else hipLaunchKernelGGL(read_lds_kernel, grows, dim3(256), 0, 0, S, M, N, tchunk, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 4 && ms < best) best = ms;
            }
            const double b = mode == 0 ? 4.0 * M * (N + 1) : bytes;
            printf("%-9s grid %5d: %7.3f ms  %6.2f TB/s  (%.0f %% of 8 TB/s)\n", mode == 0 ? "linear" : mode == 1 ? "rows" : "rows_lds",
                   mode == 0 ? grid0 : (int)(grows.x * grows.y), best, b / best / 1e9, b / best / 1e9 / 8.0 * 100);
        }
    }
    run_walk<1, 8, false>(S, M, N, sink, e0, e1);   // lsm_pass2_kernel's pattern
    run_walk<1, 8, true>(S, M, N, sink, e0, e1);
    run_walk<1, 16, false>(S, M, N, sink, e0, e1);
    run_walk<2, 8, false>(S, M, N, sink, e0, e1);
    run_walk<2, 4, false>(S, M, N, sink, e0, e1);
    run_walk<4, 4, false>(S, M, N, sink, e0, e1);
    run_walk<4, 2, false>(S, M, N, sink, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
