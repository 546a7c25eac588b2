// Experiment: does cutting a two-pass pricing into PATH CHUNKS whose [N+1][chunk] slab fits the
// 256 MB Infinity Cache, with the generator of chunk c+1 running on one stream while pass 1 reads chunk c on another,
// beat "generator, then pass 1" over the whole matrix?  The REAL kernel bodies (omc_paths_dev.h gbm_paths_body,
// omc_lsm_dev.h lsm_pass1_body) in a bare harness; a chunk here is a column range with its own antithetic halves
// (timing only -- the product would keep the global layout).
// usage: _ubench_chunked [paths] [steps]
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I options_model_amd/csrc tools/ubench_chunked.hip -o tools/_ubench_chunked
#include "omc_lsm_dev.h"
#include "omc_paths_dev.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace omc;

template <int VEC>
__global__ __launch_bounds__(kBlock) void gen(PathArgs g) { gbm_paths_body<VEC, true>(g); }
__global__ __launch_bounds__(kBlock) void p1(Pass1Args a) { lsm_pass1_body<4, 4, 1>(a); }

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

struct Setup {
    float* S;
    int64_t ld, M;
    int N;
    double* D;
    double* part1;
};

static void launch_gen(hipStream_t st, const Setup& s, int64_t col0, int64_t Mc, int vec)
{
    const double dt = 1.0 / s.N, L2E = 1.4426950408889634074;
    PathArgs g{};
    g.S = s.S + col0; g.ld = s.ld; g.P = Mc / 2; g.n_steps = s.N; g.s_init = 100.0f;
    g.a = (float)((0.05 - 0.5 * 0.04) * dt * L2E);
    g.b = (float)(0.2 * sqrt(dt) * L2E);
    g.k0 = 42; g.k1 = 0; g.stream = 0; g.pair_offset = (uint64_t)(col0 / 2);
    const int64_t items = (g.P + vec - 1) / vec;
    const dim3 grid((unsigned)((items + kBlock - 1) / kBlock));
    if (vec == 4) hipLaunchKernelGGL(gen<4>, grid, dim3(kBlock), 0, st, g);
    else if (vec == 2) hipLaunchKernelGGL(gen<2>, grid, dim3(kBlock), 0, st, g);
    else hipLaunchKernelGGL(gen<1>, grid, dim3(kBlock), 0, st, g);
}

static void launch_p1(hipStream_t st, const Setup& s, int64_t col0, int64_t Mc, int tchunk, double* part)
{
    Pass1Args a;
    a.S = s.S + col0; a.ld = s.ld; a.M = Mc; a.N = s.N; a.is_put = 1; a.K = 100.0; a.invK = 0.01; a.D = s.D;
    a.ntiles = (Mc + 1023) / 1024;
    a.part1 = part; a.tchunk = tchunk;
    const dim3 grid((unsigned)((a.ntiles + 3) / 4), (unsigned)((s.N - 1 + tchunk - 1) / tchunk));
    hipLaunchKernelGGL(p1, grid, dim3(kBlock), 0, st, a);
}

int main(int argc, char** argv)
{
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int N = argc > 2 ? atoi(argv[2]) : 252;
    Setup s;
    s.M = M; s.N = N; s.ld = (M + 63) / 64 * 64;
    CK(hipMalloc(&s.S, sizeof(float) * (size_t)s.ld * (N + 1)));
    std::vector<double> hd(N + 1);
    for (int k = 0; k <= N; ++k) hd[k] = exp(-0.05 / N * k);
    CK(hipMalloc(&s.D, sizeof(double) * (N + 1)));
    CK(hipMemcpy(s.D, hd.data(), sizeof(double) * (N + 1), hipMemcpyHostToDevice));
    const int64_t ntiles = (M + 1023) / 1024 + 64;
    CK(hipMalloc(&s.part1, sizeof(double) * 8 * (size_t)(N + 1) * (size_t)ntiles));
    hipStream_t A, A2, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&A2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t e0, e1, eB, eA2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&eB, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eA2, hipEventDisableTiming));
    std::vector<hipEvent_t> eg(64);
    for (auto& e : eg) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    auto timeit = [&](auto&& body) {
        std::vector<float> ts;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0, A));
            body();
            CK(hipEventRecord(e1, A));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 3) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        return std::make_pair(ts[0], ts[ts.size() / 2]);
    };
    auto part_of = [&](int64_t col0) { return s.part1 + 8 * (size_t)(N + 1) * (size_t)(col0 / 1024); };

    // baseline: whole matrix, one stream (the product's schedule), each generator width
    for (int vec : {4, 2, 1}) {
        auto g = timeit([&] { launch_gen(A, s, 0, M, vec); });
        printf("M=%lld N=%d  generator alone vec %d: best %.4f median %.4f ms\n", (long long)M, N, vec, g.first, g.second);
    }
    for (int tch : {84, 32}) {
        auto p = timeit([&] { launch_p1(A, s, 0, M, tch, s.part1); });
        printf("pass 1 alone (resident matrix) tchunk %d: best %.4f median %.4f ms\n", tch, p.first, p.second);
    }
    auto both = timeit([&] { launch_gen(A, s, 0, M, 4); launch_p1(A, s, 0, M, 84, s.part1); });
    printf("BASELINE generator + pass 1, one stream: best %.4f median %.4f ms\n", both.first, both.second);

    // chunked: generator chunks on A (or alternating A / A2), pass 1 of chunk c on B behind an event
    for (int C : {2, 4, 6, 8, 12, 16}) {
        int64_t Mc = (M / C + 2047) / 2048 * 2048;  // even tiles, so both halves are tile-aligned
        const int nch = (int)((M + Mc - 1) / Mc);
        for (int vec : {4, 1}) {
            for (int tch : {84, 32, 16}) {
                for (int two_gen : {0, 1}) {
                    auto r = timeit([&] {
                        if (two_gen) { CK(hipEventRecord(eA2, A)); CK(hipStreamWaitEvent(A2, eA2, 0)); }
                        CK(hipEventRecord(eB, A));
                        CK(hipStreamWaitEvent(B, eB, 0));
                        for (int c = 0; c < nch; ++c) {
                            const int64_t col0 = (int64_t)c * Mc, m = std::min(Mc, M - col0) / 2 * 2;
                            hipStream_t gs = (two_gen && (c & 1)) ? A2 : A;
                            launch_gen(gs, s, col0, m, vec);
                            CK(hipEventRecord(eg[c], gs));
                            CK(hipStreamWaitEvent(B, eg[c], 0));
                            launch_p1(B, s, col0, m, tch, part_of(col0));
                        }
                        CK(hipEventRecord(eB, B));
                        CK(hipStreamWaitEvent(A, eB, 0));
                        if (two_gen) { CK(hipEventRecord(eA2, A2)); CK(hipStreamWaitEvent(A, eA2, 0)); }
                    });
                    printf("chunks %2d (%7lld paths, %6.1f MB each) gen vec %d, pass-1 tchunk %2d, gen streams %d: best %.4f median %.4f ms\n",
                           nch, (long long)Mc, (double)Mc * 4 * (N + 1) / 1e6, vec, tch, two_gen + 1, r.first, r.second);
                }
            }
        }
    }
    // the same chunks one after the other on ONE stream (cache reuse without concurrency)
    for (int C : {4, 8, 16}) {
        int64_t Mc = (M / C + 2047) / 2048 * 2048;
        const int nch = (int)((M + Mc - 1) / Mc);
        for (int vec : {4, 1}) {
            auto r = timeit([&] {
                for (int c = 0; c < nch; ++c) {
                    const int64_t col0 = (int64_t)c * Mc, m = std::min(Mc, M - col0) / 2 * 2;
                    launch_gen(A, s, col0, m, vec);
                    launch_p1(A, s, col0, m, 32, part_of(col0));
                }
            });
            printf("serial chunks %2d gen vec %d: best %.4f median %.4f ms\n", nch, vec, r.first, r.second);
        }
    }
    CK(hipDeviceSynchronize());
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
