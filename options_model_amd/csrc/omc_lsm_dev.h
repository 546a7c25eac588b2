// omc_lsm_dev.h -- device code of the Longstaff-Schwartz backward induction (gfx950).
// Kernel bodies are __device__ functions so that the single-problem launchers (omc_lsm.hip)
// and the batched many-small-problems launchers (omc_batch.hip) share them; a body only looks
// at blockIdx.x / blockIdx.y -- the batch dimension is blockIdx.z and is resolved by the
// __global__ wrapper that picks the problem's argument struct.
#pragma once
#include "omc_device.h"
#include "omc_kernels.h"
#include <type_traits>

namespace omc {


constexpr int kPStride = kMaxLsmBlocks;  // partial-slab stride of the single-problem path

template <int VEC>
__device__ __forceinline__ void loadf(const float* __restrict__ p, float (&v)[VEC])
{
    if constexpr (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4*>(p);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
        v[0] = *p;
    }
}
// streaming variant: rows of S that this launch reads exactly once
template <int VEC>
__device__ __forceinline__ void loadf_stream(const float* __restrict__ p, float (&v)[VEC])
{
    if constexpr (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 x = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
        v[0] = __builtin_nontemporal_load(p);
    }
}
template <int VEC>
__device__ __forceinline__ void loadi(const int32_t* __restrict__ p, int32_t (&v)[VEC])
{
    if constexpr (VEC == 4) {
        const int4 x = *reinterpret_cast<const int4*>(p);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void storef(float* p, const float (&v)[VEC])
{
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *p = v[0];
}
template <int VEC>
__device__ __forceinline__ void storei(int32_t* p, const int32_t (&v)[VEC])
{
    if constexpr (VEC == 4) *reinterpret_cast<int4*>(p) = make_int4(v[0], v[1], v[2], v[3]);
    else *p = v[0];
}

// ------------------------------------------------------------------ per-step sweep
struct StepArgs {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, invK;
    float* sx;
    int32_t* tex;
    const double* D;
    double* part;
    double* gmom;
    double* betas;
    int t, nblk, external;
    int pstride;  // slots per quantity row in `part`
};

// One launch per time step t = N .. 1 (the launch boundary is the grid-wide barrier the
// regression needs).  Launch t:
//   prologue  reduce the partial moments of step t (written by launch t+1), solve beta_t
//   body      per path: apply the exercise rule at t, then add the path's contribution to
//             the moments of step t-1 -- one pass over S_t, S_{t-1} and the path state
//   epilogue  per-block partial moments of step t-1 -> part[(t-1)&1]
// SEM 0: sticky "exercised" mask (reference per-step flow).  SEM 1: textbook LSM.
//
// Geometry: 512-thread workgroups, at most 512 of them (2 per CU).  Every workgroup re-reduces
// ALL partials of the previous launch in its prologue, so their count (= the grid) is kept small:
// 512 x 64 B per workgroup is 16 MB of L2 reads per launch chip-wide, against 64 MB (6 us) with
// a 1024-workgroup grid.
constexpr int kStepBlock = 512;
constexpr int kStepWaves = kStepBlock / 64;
constexpr int kStepMaxBlocks = 512;

// 8 accumulators over the whole workgroup: per-wave LDS transpose-reduce, then threads 0..7
// add the kStepWaves wave totals of "their" quantity.  One barrier inside; returns the total
// of quantity threadIdx.x in threads 0..7.
__device__ __forceinline__ double step_block_reduce8(const double (&acc)[kNQ], double* wl, double* sh_w)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double s = wave_reduce8(acc, wl + wave * kWaveRedDoubles);
    if ((lane & 7) == 0) sh_w[wave * 8 + (lane >> 3)] = s;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < 8) {
#pragma unroll
        for (int w = 0; w < kStepWaves; ++w) tot += sh_w[w * 8 + threadIdx.x];
    }
    return tot;
}

template <int SEM, int VEC>
__device__ __forceinline__ void lsm_step_body(StepArgs a)
{
    __shared__ double wl[kStepWaves * kWaveRedDoubles];
    __shared__ double sh_w[kStepWaves * 8];
    __shared__ double sh_m[8];
    __shared__ double sh_beta[4];
    extern __shared__ double sh_D[];  // SEM 1 only: [N+1]

    const int tid = threadIdx.x;
    const int t = a.t, N = a.N;
    if ((int)blockIdx.x >= a.nblk || t > N) return;  // batched launch sized for a bigger problem
    const bool do_apply = t < N, do_mom = t >= 2, init = (t == N);

    if (SEM == 1 && do_mom) {
        for (int k = tid; k <= N; k += kStepBlock) sh_D[k] = a.D[k];
    }

    // Issue this thread's first row/state loads BEFORE the prologue: their HBM latency then
    // overlaps the partial-moment reduction and the 3x3 solve.
    const float* St = a.S + (int64_t)t * a.ld;
    const float* Sm = a.S + (int64_t)(t - 1) * a.ld;
    const int64_t stride = (int64_t)a.nblk * kStepBlock * VEC;
    int64_t j = ((int64_t)blockIdx.x * kStepBlock + tid) * VEC;
    float st[VEC], sm[VEC], sx[VEC];
    int32_t tex[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { st[v] = sm[v] = sx[v] = 0.f; tex[v] = 0; }
    auto load_chunk = [&](int64_t jj) {
        loadf<VEC>(St + jj, st);
        if (do_mom) loadf<VEC>(Sm + jj, sm);
        if (!init) {
            loadf<VEC>(a.sx + jj, sx);
            loadi<VEC>(a.tex + jj, tex);
        }
    };
    if (j < a.M) load_chunk(j);

    double b0 = 0.0, b1 = 0.0, b2 = 0.0, nfit = 0.0;
    if (do_apply) {
        if (a.external) {
            if (tid < 8) sh_m[tid] = a.gmom[(size_t)t * 8 + tid];
        } else {
            double acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.0;
            const double* pp = a.part + (size_t)(t & 1) * 8 * a.pstride;
            if (tid < a.nblk) {  // nblk <= kStepMaxBlocks == kStepBlock: one partial per thread
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = pp[q * a.pstride + tid];
            }
            const double s = step_block_reduce8(acc, wl, sh_w);
            if (tid < 8) sh_m[tid] = s;
        }
        __syncthreads();
        if (tid == 0) {
            double m[8], beta[3];
#pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = sh_m[q];
            solve_poly2(m, beta);
            sh_beta[0] = beta[0]; sh_beta[1] = beta[1]; sh_beta[2] = beta[2]; sh_beta[3] = m[0];
            if (blockIdx.x == 0) {
                double* bo = a.betas + (size_t)t * 4;
                bo[0] = beta[0]; bo[1] = beta[1]; bo[2] = beta[2]; bo[3] = m[0];
                if (!a.external) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a.gmom[(size_t)t * 8 + q] = m[q];
                }
            }
        }
        __syncthreads();
        b0 = sh_beta[0]; b1 = sh_beta[1]; b2 = sh_beta[2]; nfit = sh_beta[3];
    } else if (SEM == 1 && do_mom) {
        __syncthreads();
    }

    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const double Dm = do_mom ? a.D[N - (t - 1)] : 0.0;
    const double K = a.K, invK = a.invK;
    const int is_put = a.is_put;
    const bool fit_ok = do_apply && nfit > 0.5;
    while (j < a.M) {
        bool changed = false;
        if (init) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { sx[v] = st[v]; tex[v] = N; }
            changed = true;
        }
        if (fit_ok) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double imm = payoff_d(st[v], K, is_put);
                if (imm > 0.0 && (SEM == 1 || tex[v] == N)) {
                    const double u = fma((double)st[v], invK, -1.0);
                    const double cont = fma(u, fma(u, b2, b1), b0);
                    if (imm > cont) { sx[v] = st[v]; tex[v] = t; changed = true; }
                }
            }
        }
        if (changed) {
            storef<VEC>(a.sx + j, sx);
            storei<VEC>(a.tex + j, tex);
        }
        if (do_mom) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double imm = payoff_d(sm[v], K, is_put);
                if (imm > 0.0 && (SEM == 1 || tex[v] == N)) {
                    double p = payoff_d(sx[v], K, is_put);
                    p = p > 0.0 ? p : 0.0;
                    const double y = p * (SEM == 1 ? sh_D[tex[v] - (t - 1)] : Dm);
                    accumulate_moments(acc, fma((double)sm[v], invK, -1.0), y);
                }
            }
        }
        j += stride;
        if (j < a.M) load_chunk(j);
    }
    if (do_mom) {
        const double s = step_block_reduce8(acc, wl, sh_w);
        if (tid < 8)
            a.part[(size_t)((t - 1) & 1) * 8 * a.pstride + (size_t)tid * a.pstride + blockIdx.x] = s;
    }
}

// partial moments of step t -> gmom[t] (used when the moments leave the GPU between steps)
__device__ __forceinline__ void lsm_reduce_step_body(const double* __restrict__ part, double* __restrict__ gmom,
                                                     int t, int nblk, int pstride)
{
    __shared__ double red[kNQ * kRedStride];
    const int tid = threadIdx.x;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const double* pp = part + (size_t)(t & 1) * 8 * pstride;
    for (int i = tid; i < nblk; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += pp[q * pstride + i];
    }
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) gmom[(size_t)t * 8 + (tid >> 3)] = s;
}

// ------------------------------------------------------------------ two-pass flow
// Float32 threshold of the in-the-money test: for a float32 price s and a float64 strike K,
// K - (double)s > 0  <=>  s < thr (put), (double)s - K > 0  <=>  s > thr (call), with thr the
// nearest float32 at or above K (put) / at or below K (call).
__device__ __forceinline__ float next_float(float f, bool up)
{
    if (f == 0.0f) return __uint_as_float(up ? 0x00000001u : 0x80000001u);
    const uint32_t b = __float_as_uint(f);
    return __uint_as_float(((f > 0.0f) == up) ? b + 1u : b - 1u);
}
__device__ __forceinline__ float itm_threshold(double K, int is_put)
{
    const float Kf = (float)K;
    if (is_put) return (double)Kf < K ? next_float(Kf, true) : Kf;
    return (double)Kf > K ? next_float(Kf, false) : Kf;
}

struct Pass1Args {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, invK;
    const double* D;
    double* part1;
    int64_t ntiles;
    int tchunk;
};

// Pass 1 (options_model_3.py:482-516): no decisions, so every time step is independent.
// Work item = one WAVE x (TPW tiles of 64*VEC paths) x (a chunk of time steps); waves never
// meet at a workgroup barrier.  Per step a lane folds TPW*VEC paths into its 8 accumulators,
// the wave reduces them through its private LDS patch and writes one partial per quantity.
// The next step's rows are loaded before the current one is reduced.
// Targets are the discounted TERMINAL payoffs.
// PUT: 1 put, 0 call, -1 decided at run time (the batched launch mixes both).
template <int VEC, int TPW, int PUT = -1>
__device__ __forceinline__ void lsm_pass1_body(Pass1Args a)
{
    __shared__ double wl[kBlock / 64][kWaveRedDoubles];
    __shared__ double shD[kBlock / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tg = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    if (tg >= a.ntiles) return;  // whole wave leaves; no workgroup barrier below
    const int64_t base = tg * (64 * VEC * TPW) + (int64_t)lane * VEC;
    const int t0 = 1 + blockIdx.y * a.tchunk;
    const int t1 = min(t0 + a.tchunk, a.N);
    if (t0 >= t1) return;
    const double K = a.K, invK = a.invK;
    const int is_put = PUT < 0 ? a.is_put : PUT;
    // The chunk's discount factors go through the wave's LDS patch: a vector-memory load of
    // D[N-t] inside the loop would sit behind the row prefetch in the in-order vmcnt queue
    // and expose the prefetch latency every step.
    if (lane < t1 - t0) shD[wave][lane] = a.D[a.N - (t0 + lane)];
    // Padding columns (beyond M) read column 0 and are masked out: every load below is
    // unconditional, so the compiler can count outstanding loads instead of draining them.
    const float* colp[TPW];
    double pN[TPW][VEC];
    bool valid[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int64_t j = base + (int64_t)k * 64 * VEC;
        valid[k] = j < a.M;
        colp[k] = a.S + (valid[k] ? j : 0);
        float sn[VEC];
        loadf<VEC>(colp[k] + (int64_t)a.N * a.ld, sn);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const double p = payoff_d(sn[v], K, is_put);
            pN[k][v] = (valid[k] && p > 0.0) ? p : 0.0;
        }
    }
    auto load_rows = [&](float (&buf)[TPW][VEC], int t) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) loadf_stream<VEC>(colp[k] + (int64_t)t * a.ld, buf[k]);
    };
    // Branch-free accumulation: an out-of-the-money (or padding) path contributes u = 0, p = 0
    // (adding +0.0 is exact, so the sums are those of the masked loop, bit for bit), and the
    // step's discount factor multiplies the three target sums once per lane instead of once per
    // path.  In the money means K - S > 0 (put) / S - K > 0 (call) in float64, which for a
    // float32 S is exactly S < thr / S > thr against a float32 threshold: a 32-bit compare
    // replaces a float64 fma + compare.  The set size is only needed per wave, so it is counted
    // on the scalar unit (popcount of the compare mask).
    const float thr = itm_threshold(K, is_put);
    float thrk[TPW];  // padding tiles: a threshold no price passes
#pragma unroll
    for (int k = 0; k < TPW; ++k) thrk[k] = valid[k] ? thr : (is_put ? -__builtin_inff() : __builtin_inff());
    auto process = [&](auto put_tag, const float (&buf)[TPW][VEC], int t) {
        constexpr bool IS_PUT = decltype(put_tag)::value;
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.0;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float sf = buf[k][v];
                const bool itm = IS_PUT ? sf < thrk[k] : sf > thrk[k];
                cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(itm));
                const double u0 = fma((double)sf, invK, -1.0);
                const double u = itm ? u0 : 0.0;
                const double p = itm ? pN[k][v] : 0.0;
                const double u2 = u * u;
                acc[1] += u;
                acc[2] += u2;
                acc[3] = fma(u2, u, acc[3]);
                acc[4] = fma(u2, u2, acc[4]);
                acc[5] += p;
                acc[6] = fma(u, p, acc[6]);
                acc[7] = fma(u2, p, acc[7]);
            }
        }
        const double d = shD[wave][t - t0];
        acc[0] = lane == 0 ? (double)cnt : 0.0;
        acc[5] *= d;
        acc[6] *= d;
        acc[7] *= d;
        const double s = wave_reduce8(acc, wl[wave]);
        if ((lane & 7) == 0) a.part1[((size_t)t * 8 + (lane >> 3)) * a.ntiles + tg] = s;
    };
    // Rows are read with the nontemporal hint (each byte is used once per launch): 253 -> 218 us at
    // C2; the same hint on pass 2 or on the generator's stores did nothing or lost a little.
    // Three rotating register buffers, rows fetched TWO steps ahead of their use; row indices are
    // clamped to the chunk, so every load is unconditional and the compiler can count them
    // (vmcnt(7..4) in the ISA instead of vmcnt(0)).  Measured (SQ counters):
    // wave-cycles split 46 % VALU issue stall / 35 % memory wait / 19 % issuing.  Loads alone or
    // arithmetic alone each take ~0.20 ms at C2, together ~0.27 ms; neither a deeper prefetch
    // nor 15-20 % fewer VALU instructions (float32 threshold compare, scalar popcount) moved it.
    float bufA[TPW][VEC], bufB[TPW][VEC], bufC[TPW][VEC];
    const int tl = t1 - 1;
    // sched_barrier: hipcc otherwise sinks the prefetch loads below the arithmetic they are
    // meant to overlap (seen in the ISA as vmcnt(0) right before the late-issued loads)
    auto sweep = [&](auto put_tag) {
        load_rows(bufA, t0);
        load_rows(bufB, min(t0 + 1, tl));
        for (int t = t0; t < t1; t += 3) {
            load_rows(bufC, min(t + 2, tl));
            __builtin_amdgcn_sched_barrier(0);
            process(put_tag, bufA, t);
            if (t + 1 < t1) {
                load_rows(bufA, min(t + 3, tl));
                __builtin_amdgcn_sched_barrier(0);
                process(put_tag, bufB, t + 1);
            }
            if (t + 2 < t1) {
                load_rows(bufB, min(t + 4, tl));
                __builtin_amdgcn_sched_barrier(0);
                process(put_tag, bufC, t + 2);
            }
        }
    };
    if (PUT == 1 || (PUT < 0 && is_put))
        sweep(std::true_type{});
    else
        sweep(std::false_type{});
}

__device__ __forceinline__ void lsm_reduce_pass1_body(const double* __restrict__ part1,
                                                      double* __restrict__ gmom, int64_t ntiles,
                                                      int N)
{
    if ((int)blockIdx.x + 1 >= N) return;  // batched launches are sized for the longest problem
    // one workgroup per (step, quantity): 8x more workgroups than one per step, each with a
    // short strided sum -- the slab read is latency-bound, so parallelism is what it needs
    __shared__ double sh[kBlock / 64];
    const int tid = threadIdx.x;
    const int t = blockIdx.x + 1, q = blockIdx.y;
    const double* pp = part1 + ((size_t)t * 8 + q) * ntiles;
    double s = 0.0;
    for (int64_t i = tid; i < ntiles; i += kBlock) s += pp[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) gmom[(size_t)t * 8 + q] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ void lsm_solve_all_body(const double* __restrict__ gmom, double* __restrict__ betas,
                                     int N)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 1 || t >= N) return;
    double m[8], beta[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = gmom[(size_t)t * 8 + q];
    solve_poly2(m, beta);
    double* bo = betas + (size_t)t * 4;
    bo[0] = beta[0]; bo[1] = beta[1]; bo[2] = beta[2]; bo[3] = m[0];
}

struct Pass2Args {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, invK;
    const double* D;
    const double* betas;
    float* sx;
    int32_t* tex;
    double* part;
    int nblk, pstride;
};

// Pass 2 (options_model_3.py:615-651) with frozen per-step fits: every path is
// independent, so one thread walks its VEC paths backward through all steps and stops as
// soon as they have all exercised (sticky mask).  Sums of the t=dt-valued cash-flows are
// reduced per block.
template <int VEC, bool WRITE_STATE>
__device__ __forceinline__ void lsm_pass2_body(Pass2Args a)
{
    if ((int)blockIdx.x >= a.nblk) return;
    __shared__ double red[kNQ * kRedStride];
    extern __shared__ double sh_b[];  // [N+1][4]
    const int tid = threadIdx.x;
    const int N = a.N;
    // a step with an empty regression set never exercises: give it an infinite continuation
    // value instead of a branch in the sweep
    for (int k = tid; k < (N + 1) * 4; k += kBlock) {
        const int t = k >> 2;
        const bool fit = t >= 1 && t < N && a.betas[(size_t)t * 4 + 3] > 0.5;
        sh_b[k] = fit ? a.betas[k] : ((k & 3) == 0 ? __builtin_huge_val() : 0.0);
    }
    __syncthreads();
    const double K = a.K, invK = a.invK;
    const int is_put = a.is_put;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)a.nblk * kBlock * VEC;
    for (int64_t j = ((int64_t)blockIdx.x * kBlock + tid) * VEC; j < a.M; j += stride) {
        float sx[VEC];
        int32_t tex[VEC];
        loadf<VEC>(a.S + (int64_t)N * a.ld + j, sx);
#pragma unroll
        for (int v = 0; v < VEC; ++v) tex[v] = N;
        // One row of decisions, branch-free: a path that has already exercised (tex != N), is
        // out of the money, or sits below the fitted continuation value keeps its state.
        auto decide = [&](const float (&row)[VEC], int t) {
            const double b0 = sh_b[4 * t], b1 = sh_b[4 * t + 1], b2 = sh_b[4 * t + 2];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double sd = (double)row[v];
                const double imm = is_put ? K - sd : sd - K;
                const double u = fma(sd, invK, -1.0);
                const double cont = fma(u, fma(u, b2, b1), b0);
                const bool ex = (tex[v] == N) & (imm > 0.0) & (imm > cont);
                sx[v] = ex ? row[v] : sx[v];
                tex[v] = ex ? t : tex[v];
            }
        };
        auto live = [&]() {
            bool l = false;
#pragma unroll
            for (int v = 0; v < VEC; ++v) l |= (tex[v] == N);
            return l;
        };
        constexpr int U = 8;  // full blocks of U rows: U unconditional 16-byte loads in flight per lane
        int t = N - 1;
        const float* col = a.S + j;
        for (; t >= U && live(); t -= U) {
            float st[U][VEC];
#pragma unroll
            for (int k = 0; k < U; ++k) loadf<VEC>(col + (int64_t)(t - k) * a.ld, st[k]);
#pragma unroll
            for (int k = 0; k < U; ++k) decide(st[k], t - k);
        }
        for (; t >= 1 && live(); --t) {
            float st[VEC];
            loadf<VEC>(col + (int64_t)t * a.ld, st);
            decide(st, t);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            double p = payoff_d(sx[v], K, is_put);
            p = p > 0.0 ? p : 0.0;
            const double cf = p * a.D[tex[v] - 1];
            acc[0] += cf;
            acc[1] += cf * cf;
            acc[2] += (tex[v] < N) ? 1.0 : 0.0;
            acc[3] += (cf == 0.0) ? 1.0 : 0.0;
        }
        if (WRITE_STATE) {
            storef<VEC>(a.sx + j, sx);
            storei<VEC>(a.tex + j, tex);
        }
    }
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) a.part[(size_t)(tid >> 3) * a.pstride + blockIdx.x] = s;
}

// ------------------------------------------------------------------ valuation + finalize
struct FinalArgs {
    const float* sx;
    const int32_t* tex;
    int64_t M;
    int N, is_put, tval;
    double K;
    const double* D;
    double* part;
    int nblk, pstride;
};

template <int VEC>
__device__ __forceinline__ void lsm_final_body(FinalArgs a)
{
    if ((int)blockIdx.x >= a.nblk) return;
    __shared__ double red[kNQ * kRedStride];
    const int tid = threadIdx.x;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)a.nblk * kBlock * VEC;
    for (int64_t j = ((int64_t)blockIdx.x * kBlock + tid) * VEC; j < a.M; j += stride) {
        float sx[VEC];
        int32_t tex[VEC];
        loadf<VEC>(a.sx + j, sx);
        loadi<VEC>(a.tex + j, tex);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            double p = payoff_d(sx[v], a.K, a.is_put);
            p = p > 0.0 ? p : 0.0;
            const double cf = p * a.D[tex[v] - a.tval];
            acc[0] += cf;
            acc[1] += cf * cf;
            acc[2] += (tex[v] < a.N) ? 1.0 : 0.0;
            acc[3] += (cf == 0.0) ? 1.0 : 0.0;
        }
    }
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) a.part[(size_t)(tid >> 3) * a.pstride + blockIdx.x] = s;
}

// part[0][q][0..nblk) -> result[q]; result[4] = sum over t of the regression-set sizes
__device__ __forceinline__ void lsm_finalize_body(const double* __restrict__ part,
                                                  const double* __restrict__ gmom,
                                                  double* __restrict__ result, int nblk, int N,
                                                  int pstride)
{
    __shared__ double red[kNQ * kRedStride];
    const int tid = threadIdx.x;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int i = tid; i < nblk; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += part[(size_t)q * pstride + i];
    }
    for (int t = 1 + tid; t < N; t += kBlock) acc[4] += gmom[(size_t)t * 8];
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) result[tid >> 3] = s;
}

}  // namespace omc
