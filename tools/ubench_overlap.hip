// How well do a streaming row read and float64 arithmetic on the loaded values overlap on gfx950, as a function of
// the arithmetic per value and of the waves per SIMD?  Same tiling as lsm_pass1_kernel: a wave owns 1024 consecutive
// paths, walks 32 rows 4 MB apart, three rotating register buffers (rows fetched two steps ahead), F float64 FMAs per
// loaded value into 8 accumulator chains.  Prints time against max(load-only, arithmetic-only) and their sum.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I options_model_amd/csrc tools/ubench_overlap.hip -o tools/_ubench_overlap
#include <hip/hip_runtime.h>
#include "omc_device.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef float f4 __attribute__((ext_vector_type(4)));

// F: FMAs per value (F < 0: the moment arithmetic of lsm_pass1_kernel; bit 0 of -F: mask by compare + select,
// bit 1: in-the-money count through ballot + s_bcnt1, bit 2: per-step wave reduce through LDS + 8 stores);
// LOADS: 1 = real loads, 0 = rows loaded once (arithmetic only); PAD_KB limits workgroups per CU
template <int F, int LOADS, int PAD_KB>
__global__ __launch_bounds__(256) void k(const float* __restrict__ S, int64_t M, int N, int tchunk, double* sink)
{
    __shared__ float pad[PAD_KB > 0 ? PAD_KB * 256 : 1];
    if (PAD_KB > 0 && M < 0) pad[threadIdx.x] = 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntiles = (M + 1023) / 1024;
    const int64_t tg = (int64_t)blockIdx.x * 4 + wave;
    if (tg >= ntiles) return;
    const int t0 = 1 + blockIdx.y * tchunk, t1 = min(t0 + tchunk, N);
    const float* col = S + tg * 1024 + lane * 4;
    f4 a[4], b[4], c[4];
    bool loaded = false;
    auto ld = [&](f4 (&d)[4], int t) {
        if (!LOADS && loaded) {
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(d[q]));
            return;
        }
        t = min(t, t1 - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(col + (int64_t)t * M + q * 256));
    };
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t fold = 0;
    __shared__ double red[4][8 * 65];
    __shared__ double wlr[4][omc::kWaveRedDoubles];
    double pN[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pN[i] = 1.0 + 0.001 * (lane + i);
    const double invK = 0.01;
    const float thr = 1.5f;
    int step = 0;
    auto eat_real = [&](const f4 (&d)[4]) {
        constexpr int OPT = -F;
        double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {d[q].x, d[q].y, d[q].z, d[q].w};
            double u[4], m[4], u2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = __builtin_fma((double)v[e], invK, -1.0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (OPT & 1) {
                    const bool itm = v[e] < thr;
                    if (OPT & 2) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(itm));
                    m[e] = itm ? 1.0 : 0.0;
                } else {
                    m[e] = 1.0;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] *= m[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) u2[e] = u[e] * u[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a8[1] += u[e];
                a8[2] += u2[e];
                a8[3] = __builtin_fma(u2[e], u[e], a8[3]);
                a8[4] = __builtin_fma(u2[e], u2[e], a8[4]);
                a8[5] = __builtin_fma(pN[4 * q + e], m[e], a8[5]);
                a8[6] = __builtin_fma(u[e], pN[4 * q + e], a8[6]);
                a8[7] = __builtin_fma(u2[e], pN[4 * q + e], a8[7]);
            }
        }
        a8[0] = (double)cnt;
        if (OPT & 4) {
            // transpose-reduce through the wave's LDS patch, 8 lanes store (as wave_reduce8 does in spirit)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[wave][i * 65 + lane] = a8[i];
            double s2 = 0.0;
            const int qq = lane >> 3, part = lane & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) s2 += red[wave][qq * 65 + part * 8 + i];
            if (OPT & 16) {  // the three levels inside groups of 8 lanes through DPP moves instead of ds_bpermute
                auto dpp_add = [&](double x, auto ctrl) {
                    constexpr int C = decltype(ctrl)::value;
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), C, 0xf, 0xf, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), C, 0xf, 0xf, false);
                    return x + __hiloint2double(hi, lo);
                };
                s2 = dpp_add(s2, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]: lane ^ 1
                s2 = dpp_add(s2, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]: lane ^ 2
                s2 = dpp_add(s2, std::integral_constant<int, 0x141>{});  // row_half_mirror: the other quad of the 8
            } else {
                s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
            }
            if (OPT & 64) { acc[0] += omc::wave_reduce8(a8, wlr[wave]); }  // the library's own wave reduction instead
            else if (OPT & 32) acc[0] += s2;  // reduce, but nothing leaves the wave inside the loop
            else if (part == 0) sink[512 + ((size_t)(blockIdx.y * 64 + step) * 8 + qq) * 4096 + (tg & 4095)] = s2;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += a8[i];
        }
        ++step;
    };
    auto eat = [&](const f4 (&d)[4]) {
        if constexpr (F < 0) { eat_real(d); return; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {d[q].x, d[q].y, d[q].z, d[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (F == 0) {
                    fold ^= __float_as_uint(v[e]);
                } else {
                    const double u = (double)v[e];
#pragma unroll
                    for (int f = 0; f < F; ++f) acc[f & 7] = __builtin_fma(u, acc[(f + 1) & 7] + 1e-30 * (f >= 8), acc[f & 7]);
                }
            }
        }
    };
    ld(a, t0); ld(b, t0 + 1);
    if (!LOADS) { ld(c, t0 + 2); loaded = true; }
    for (int t = t0; t < t1; t += 3) {
        ld(c, t + 2); __builtin_amdgcn_sched_barrier(0); eat(a);
        ld(a, t + 3); __builtin_amdgcn_sched_barrier(0); eat(b);
        ld(b, t + 4); __builtin_amdgcn_sched_barrier(0); eat(c);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 1.2345 || fold == 0x12345678u) sink[threadIdx.x] = s;
}

template <int F, int LOADS, int PAD_KB>
static float run(const float* S, int64_t M, int N, double* sink)
{
    const int tchunk = 33;
    const dim3 grid((unsigned)((M / 1024 + 1 + 3) / 4), (N - 1 + tchunk - 1) / tchunk);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<F, LOADS, PAD_KB>), grid, dim3(256), 0, 0, S, M, N, tchunk, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    return best;
}

template <int F, int PAD_KB>
static void row(const char* occ, const float* S, int64_t M, int N, double* sink, float load_only)
{
    const float both = run<F, 1, PAD_KB>(S, M, N, sink), arith = run<F, 0, PAD_KB>(S, M, N, sink);
    const float mx = load_only > arith ? load_only : arith;
    printf("%s  F=%2d: loads %.3f  arithmetic %.3f  both %.3f ms   = %.2f x max, %.2f x sum\n", occ, F, load_only, arith, both,
           both / mx, both / (load_only + arith));
}

template <int PAD_KB>
static void table(const char* occ, const float* S, int64_t M, int N, double* sink)
{
    const float lo = run<0, 1, PAD_KB>(S, M, N, sink);
    row<8, PAD_KB>(occ, S, M, N, sink, lo);
    row<-8, PAD_KB>(occ, S, M, N, sink, lo);   // moments, no mask
    row<-1, PAD_KB>(occ, S, M, N, sink, lo);   // + compare / select mask
    row<-3, PAD_KB>(occ, S, M, N, sink, lo);   // + ballot count
    row<-7, PAD_KB>(occ, S, M, N, sink, lo);   // + per-step reduce and stores
    row<-23, PAD_KB>(occ, S, M, N, sink, lo);  // the same with DPP moves for the last three levels
    row<-39, PAD_KB>(occ, S, M, N, sink, lo);  // reduce (bpermute) but no store in the loop
    row<-55, PAD_KB>(occ, S, M, N, sink, lo);  // reduce (DPP) but no store in the loop
    row<-71, PAD_KB>(occ, S, M, N, sink, lo);  // omc::wave_reduce8, no store in the loop
}

int main(int argc, char** argv)
{
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int N = 252;
    float* S;
    double* sink;
    hipMalloc(&S, sizeof(float) * M * (N + 1));
    hipMalloc(&sink, sizeof(double) * (512 + (size_t)8 * 64 * 8 * 4096));
    hipMemset(S, 0x3f, sizeof(float) * M * (N + 1));
    table<40>("<=4 waves/SIMD", S, M, N, sink);
    table<53>("<=3 waves/SIMD", S, M, N, sink);
    table<80>("<=2 waves/SIMD", S, M, N, sink);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
