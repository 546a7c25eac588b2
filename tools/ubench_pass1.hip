// The REAL lsm_pass1_kernel body (options_model_amd/csrc/omc_lsm_dev.h) in a bare harness: constant data, no
// generator, no other kernel around it, best of several launches -- next to tools/ubench_overlap.hip, which runs
// an imitation of the same work.  usage: _ubench_pass1 [paths]
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I options_model_amd/csrc tools/ubench_pass1.hip -o tools/_ubench_pass1
#include "omc_lsm_dev.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace omc;

template <int DIAG>
__global__ __launch_bounds__(kBlock) void p1(Pass1Args a) { lsm_pass1_body<4, 4, 1, DIAG>(a); }

template <int DIAG>
static float run(const Pass1Args& a, dim3 grid)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(p1<DIAG>, grid, dim3(kBlock), 0, 0, a);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv)
{
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int N = 252;
    float* S;
    double *D, *part1;
    (void)hipMalloc(&S, sizeof(float) * M * (N + 1));
    std::vector<float> row(M);
    for (int64_t j = 0; j < M; ++j) row[j] = 80.0f + 40.0f * (float)((j * 2654435761u) % 1000) / 1000.0f;  // half in the money
    for (int t = 0; t <= N; ++t) (void)hipMemcpy(S + (size_t)t * M, row.data(), sizeof(float) * M, hipMemcpyHostToDevice);
    std::vector<double> hd(N + 1, 0.99);
    (void)hipMalloc(&D, sizeof(double) * (N + 1));
    (void)hipMemcpy(D, hd.data(), sizeof(double) * (N + 1), hipMemcpyHostToDevice);
    Pass1Args a;
    a.S = S; a.ld = M; a.M = M; a.N = N; a.is_put = 1; a.K = 100.0; a.invK = 0.01; a.D = D;
    a.ntiles = (M + 1023) / 1024;
    (void)hipMalloc(&part1, sizeof(double) * 8 * (N + 1) * a.ntiles);
    a.part1 = part1;
    for (int tch : {33, 63, 15}) {
        a.tchunk = tch;
        const dim3 grid((unsigned)((a.ntiles + 3) / 4), (unsigned)((N - 1 + tch - 1) / tch));
        printf("M=%lld tchunk %2d: full %.4f  arithmetic only %.4f  loads only %.4f  no reduce %.4f ms\n", (long long)M, tch,
               run<0>(a, grid), run<1>(a, grid), run<2>(a, grid), run<3>(a, grid));
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
