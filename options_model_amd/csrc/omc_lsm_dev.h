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
    } else if constexpr (VEC == 2) {
        const float2 x = *reinterpret_cast<const float2*>(p);
        v[0] = x.x; v[1] = x.y;
    } else {
        static_assert(VEC == 1, "loadf: 1, 2 or 4 floats");
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
    } else if constexpr (VEC == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 x = __builtin_nontemporal_load(reinterpret_cast<const f2*>(p));
        v[0] = x.x; v[1] = x.y;
    } else {
        static_assert(VEC == 1, "loadf_stream: 1, 2 or 4 floats");
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
    float* live;  // SEM 0: the path's TERMINAL spot S_N while it has not exercised, a negative value once it has
                  // (sticky; sx / tex are then only WRITTEN, at exercise): state and regression target in one row
    const double* D;
    double* part;
    double* gmom;
    double* betas;
    int t, nblk, external;
    int pstride;  // slots per quantity row in `part`
    int gstride;  // doubles between consecutive steps' rows of `gmom` (8; K * 8 when K pricings share one table)
    // "values" mode (omc_lsm_apply_values): the continuation value of path j at step t is cont[t][j]
    // (float32, as the reference's networks return it) instead of the fitted polynomial
    const float* cont;
    int64_t ldc;
};

// One launch per time step t = N .. 1 (the launch boundary is the grid-wide barrier the
// regression needs: step t's regression set depends on every path's decision at t+1).  Launch t:
//   prologue  wave 0 reduces the per-workgroup partial moments of step t (written by launch t+1),
//             solves beta_t and hands it to the workgroup through LDS -- ONE barrier; the other
//             waves already have their row / state loads in flight
//   body      per path: apply the exercise rule at t, then add the path's contribution to
//             the moments of step t-1 -- one pass over S_t, S_{t-1} and the path state
//   epilogue  per-workgroup partial moments of step t-1 -> part[(t-1)&1]
// SEM 0: sticky "exercised" mask (reference per-step flow, Options_model.py:108-150).  A path in the
//        regression set has by construction never exercised, so its cash-flow is the discounted
//        TERMINAL payoff: the loop reads S_t, S_{t-1} and the path's `live` value -- S_N while it is in
//        play, negative once it has exercised -- i.e. 12 bytes per path and step (round 2 read S_N from the
//        matrix plus a flag byte: 13), and writes (sx, tex, live) only when a path exercises -- once per path.
// SEM 1: textbook LSM: the cash-flow of every in-the-money path is payoff(sx) D[tex - t]: state
//        (sx, tex) is read every step (16 bytes per path and step).
//
// Geometry.  The paths of a pricing are cut into a.nblk SLOTS (at most kStepMaxBlocks): slot b owns the
// column chunks (b*BLOCK + tid)*VEC + i * nblk*BLOCK*VEC, and its partial moments -- thread sums over its
// chunks in column order, wave transpose-reduce, waves added in wave order -- go to part[..][b].  That
// summation tree depends on (M, BLOCK, VEC) only, NOT on how many workgroups run it:
//   * one pricing per launch: G = nblk workgroups, one slot each (w = blockIdx.x);
//   * K pricings per launch (lsm_step_multi_kernel): each pricing gets G = workgroups / K of the chip's
//     workgroups and workgroup w of a pricing walks slots w, w + G, w + 2G, ...  -- so the K pricings are
//     co-resident and share ONE launch boundary and ONE cold start per time step, and each of them returns the
//     bits of its own single launch.  (Consecutive, XCD-aware runs of slots instead -- what pass 1 gains 3.5 % from
//     -- change nothing here: 0.758-0.763 against 0.760-0.764 ms per pricing at K = 16, round 4.)
// Workgroup count per pricing = number of partials every workgroup's wave 0 re-reads in its prologue
// (kStepMaxBlocks / 64 per lane and quantity, all issued before anything waits).
constexpr int kStepMaxBlocks = 256;
constexpr int kStepPL = kStepMaxBlocks / 64;
constexpr int kStepMaxItems = 32;  // slots one workgroup walks at most (K <= 32 pricings per launch)

template <int SEM, int VEC, int BLOCK>
__device__ __forceinline__ void lsm_step_body(StepArgs a, const int w, const int G)
{
    constexpr int WAVES = BLOCK / 64;
    __shared__ double wl[WAVES * kWaveRedDoubles];
    __shared__ double sh_w[kStepMaxItems * WAVES * 8];
    __shared__ double sh_beta[4];
    extern __shared__ double sh_D[];  // SEM 1 only: [N+1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = a.t, N = a.N;
    if (w >= a.nblk || t > N) return;  // batched launch sized for a bigger problem
    const bool do_apply = t < N, do_mom = t >= 2, init = (t == N);
    const bool values = a.cont != nullptr;
    const float* controw = values ? a.cont + (int64_t)t * a.ldc : nullptr;

    if (SEM == 1 && do_mom) {
        for (int k = tid; k <= N; k += BLOCK) sh_D[k] = a.D[k];
    }

    // ---- prologue loads (wave 0): the partial moments of step t, issued FIRST so that their
    // counted wait (vmcnt) does not sit behind this wave's own row loads.  (Measured with the STAMP
    // build at C2, DESIGN.md section 8: the first bytes of ANY load come back 1.6-2.3 us after kernel
    // entry; holding the row loads back until the partials are in, or de-correlating the order in which
    // workgroups walk the partial rows, changed nothing measurable in the production build.)
    const bool pro = do_apply && wave == 0 && !values;
    double pv[kStepPL][8];
    if (pro && !a.external) {
        const double* pp = a.part + (size_t)(t & 1) * 8 * a.pstride;
#pragma unroll
        for (int i = 0; i < kStepPL; ++i) {
            const int b = lane + 64 * i;
            const bool ok = b < a.nblk;
            const double* p = pp + (ok ? b : 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double v = p[(size_t)q * a.pstride];
                pv[i][q] = ok ? v : 0.0;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the row / state chunks this thread will walk: (slot w, chunk 0), (slot w, chunk 1) ..., (slot w + G, 0) ...
    // Two chunks are kept in flight (buffers A and B, refilled right after use): the first two go out here, ahead
    // of the prologue.  With one chunk in flight per thread a workgroup has 53 KB outstanding, which caps the chip
    // at ~6.2 TB/s of the 13 bytes per path once several pricings share a launch; the second buffer costs 13 VGPRs.
    const float* St = a.S + (int64_t)t * a.ld;
    const float* Sm = St - a.ld;
    const int64_t stride = (int64_t)a.nblk * BLOCK * VEC;
    struct Chunk {
        float st[VEC], sm[VEC], sx[VEC];  // sx: SEM 0 the `live` values (S_N, or < 0: exercised), SEM 1 spot at the exercise time
        int32_t tex[VEC];                 // SEM 1 only
        uint32_t exw;                     // SEM 0: bit 8v set = path v of the chunk has exercised
        int64_t j;                        // first column of the chunk; >= M: nothing there for this lane
    };
    auto load_chunk = [&](Chunk& c) {
        loadf<VEC>(St + c.j, c.st);  // (the nontemporal hint on this row, last read here, changes nothing: measured)
        if (do_mom) loadf<VEC>(Sm + c.j, c.sm);
        if (!init) {
            if (SEM == 0) {
                loadf<VEC>(a.live + c.j, c.sx);
            } else {
                loadf<VEC>(a.sx + c.j, c.sx);
                loadi<VEC>(a.tex + c.j, c.tex);
            }
        }
    };
    // chunks of slot vb: as many as its FIRST thread has (uniform over the workgroup; lanes beyond M idle)
    auto slot_chunks = [&](int vb) { return (int)((a.M - (int64_t)vb * BLOCK * VEC + stride - 1) / stride); };
    int cur_vb = w, cur_i = 0, cur_n = slot_chunks(w);
    auto fetch = [&](Chunk& c, bool& valid, bool& last) {
        valid = cur_vb < a.nblk;
        last = false;
        if (!valid) return;
        c.j = ((int64_t)cur_vb * BLOCK + tid) * VEC + (int64_t)cur_i * stride;
        if (c.j < a.M) load_chunk(c);
        last = cur_i + 1 == cur_n;
        if (last) {
            cur_vb += G;
            cur_i = 0;
            cur_n = cur_vb < a.nblk ? slot_chunks(cur_vb) : 0;
        } else {
            ++cur_i;
        }
    };
    Chunk A, B;
    bool vA, lA, vB, lB;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        A.st[v] = A.sm[v] = A.sx[v] = B.st[v] = B.sm[v] = B.sx[v] = 0.f;
        A.tex[v] = B.tex[v] = 0;
    }
    A.exw = B.exw = 0;
    fetch(A, vA, lA);
    fetch(B, vB, lB);
    __builtin_amdgcn_sched_barrier(0);

    double b0 = 0.0, b1 = 0.0, b2 = 0.0, nfit = 0.0;
    if (do_apply && !values) {
        if (pro) {
            double m[8];
            if (a.external) {
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = a.gmom[(size_t)t * a.gstride + q];
            } else {
                double acc[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc[q] = pv[0][q];
#pragma unroll
                    for (int i = 1; i < kStepPL; ++i) acc[q] += pv[i][q];
                }
                const double s = wave_reduce8(acc, wl);  // total of quantity lane >> 3 in every lane
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = __shfl(s, 8 * q);
            }
            double beta[3];
            solve_poly2(m, beta);  // every lane, same result
            if (lane == 0) {
                sh_beta[0] = beta[0]; sh_beta[1] = beta[1]; sh_beta[2] = beta[2]; sh_beta[3] = m[0];
                if (w == 0) {
                    double* bo = a.betas + (size_t)t * 4;
                    bo[0] = beta[0]; bo[1] = beta[1]; bo[2] = beta[2]; bo[3] = m[0];
                    if (!a.external) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) a.gmom[(size_t)t * a.gstride + q] = m[q];
                    }
                }
            }
        }
        __syncthreads();
        b0 = sh_beta[0]; b1 = sh_beta[1]; b2 = sh_beta[2]; nfit = sh_beta[3];
    } else if (SEM == 1 && do_mom) {
        __syncthreads();
    }

    const double Dm = do_mom ? a.D[N - (t - 1)] : 0.0;
    const double K = a.K, invK = a.invK;
    const int is_put = a.is_put;
    const bool fit_ok = do_apply && (values || nfit > 0.5);
    int item = 0;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    bool added = false;
    // one chunk: apply the exercise rule at t, add to the moments of t-1; `last`: the slot ends here
    auto consume = [&](Chunk& c, const bool last) {
        const int64_t j = c.j;
        if (j < a.M) {
            if (SEM == 0) {
                if (init) {  // t == N: every path is in play, its live value is its terminal spot
#pragma unroll
                    for (int v = 0; v < VEC; ++v) c.sx[v] = c.st[v];
                    storef<VEC>(a.live + j, c.sx);
                }
                c.exw = 0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) c.exw |= c.sx[v] < 0.0f ? 1u << (8 * v) : 0u;
                if (fit_ok) {
                    uint32_t neww = c.exw;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double imm = payoff_d(c.st[v], K, is_put);
                        if (imm > 0.0 && ((c.exw >> (8 * v)) & 0xffu) == 0u) {
                            const double u = fma((double)c.st[v], invK, -1.0);
                            const double cont = values ? (double)controw[j + v] : fma(u, fma(u, b2, b1), b0);
                            if (imm > cont) {  // each path gets here at most once in the whole sweep
                                neww |= 1u << (8 * v);
                                a.sx[j + v] = c.st[v];
                                a.tex[j + v] = t;
                                a.live[j + v] = -1.0f;
                            }
                        }
                    }
                    c.exw = neww;
                }
                if (do_mom) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double imm = payoff_d(c.sm[v], K, is_put);
                        if (imm > 0.0 && ((c.exw >> (8 * v)) & 0xffu) == 0u) {
                            double p = payoff_d(c.sx[v], K, is_put);
                            p = p > 0.0 ? p : 0.0;
                            accumulate_moments(acc, fma((double)c.sm[v], invK, -1.0), p * Dm);
                            added = true;
                        }
                    }
                }
            } else {
                bool changed = false;
                if (init) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) { c.sx[v] = c.st[v]; c.tex[v] = N; }
                    changed = true;
                }
                if (fit_ok) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double imm = payoff_d(c.st[v], K, is_put);
                        if (imm > 0.0) {
                            const double u = fma((double)c.st[v], invK, -1.0);
                            const double cont = values ? (double)controw[j + v] : fma(u, fma(u, b2, b1), b0);
                            if (imm > cont) { c.sx[v] = c.st[v]; c.tex[v] = t; changed = true; }
                        }
                    }
                }
                if (changed) {
                    storef<VEC>(a.sx + j, c.sx);
                    storei<VEC>(a.tex + j, c.tex);
                }
                if (do_mom) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double imm = payoff_d(c.sm[v], K, is_put);
                        if (imm > 0.0) {
                            double p = payoff_d(c.sx[v], K, is_put);
                            p = p > 0.0 ? p : 0.0;
                            accumulate_moments(acc, fma((double)c.sm[v], invK, -1.0), p * sh_D[c.tex[v] - (t - 1)]);
                            added = true;
                        }
                    }
                }
            }
        }
        if (last) {  // the slot's sums: wave transpose-reduce now, waves added after the last slot
            if (do_mom) {
                // a wave none of whose lanes added anything contributes exact zeros: skip its transpose
                double s = 0.0;
                if (__builtin_amdgcn_ballot_w64(added) != 0) s = wave_reduce8(acc, wl + wave * kWaveRedDoubles);
                if ((lane & 7) == 0) sh_w[(item * WAVES + wave) * 8 + (lane >> 3)] = s;
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = 0.0;
                added = false;
            }
            ++item;
        }
    };
    for (;;) {
        if (!vA) break;
        consume(A, lA);
        fetch(A, vA, lA);
        if (!vB) break;
        consume(B, lB);
        fetch(B, vB, lB);
    }
    if (do_mom) {
        __syncthreads();
        if (tid < 8 * item) {  // one thread per (slot, quantity): waves added in wave order
            const int it = tid >> 3, q = tid & 7;
            double tot = 0.0;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) tot += sh_w[(it * WAVES + ww) * 8 + q];
            a.part[(size_t)((t - 1) & 1) * 8 * a.pstride + (size_t)q * a.pstride + (w + it * G)] = tot;
        }
    }
}

// partial moments of step t -> gmom[t] (used when the moments leave the GPU between steps)
__device__ __forceinline__ void lsm_reduce_step_body(const double* __restrict__ part, double* __restrict__ gmom,
                                                     int t, int nblk, int pstride, int gstride = 8)
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
    if (tid < 64 && (tid & 7) == 0) gmom[(size_t)t * gstride + (tid >> 3)] = s;
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

constexpr int kPass1MaxChunk = 256;  // time steps a pass-1 workgroup may take (its discount factors live in LDS)

struct Pass1Args {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, invK;
    const double* D;
    double* part1;
    int64_t ntiles;
    int tchunk;
    // antithetic-FOLDED storage (lsm_pass1_fold_body): S holds only the first partner of every pair, M = stored columns;
    // cK[t] = C_t / K with C_t = S_t S'_t = S0^2 exp(2 drift t) -- the partner's moneyness is cK[t] / S_t
    const double* cK = nullptr;
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
    __shared__ double shD[kBlock / 64][kPass1MaxChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Tile groups are numbered XCD-aware: the workgroups that run on one XCD walk one contiguous eighth of every row
    // (measured: pass 1 202 -> 195 us at C2, 1.53 -> 1.47 ms at C3's shard; the same renumbering does nothing for the
    // generator and pass 2, profiles/r04_xcd_mapping.txt).  Changes no sum: partials are per (step, tile).
    const int64_t tg = (int64_t)xcd_block((int)blockIdx.x, (int)gridDim.x) * (kBlock / 64) + wave;
    if (tg >= a.ntiles) return;  // whole wave leaves; no workgroup barrier below
    const int64_t base = tg * (64 * VEC * TPW) + (int64_t)lane * VEC;
    // Time chunks are visited LATEST FIRST (workgroups are dispatched in blockIdx order): the generator
    // has just written the whole matrix row by row for all paths at once, so what can still be in the 256 MB
    // Infinity Cache when this kernel starts are the last ~60 rows (measured effect: small, 0.198 -> 0.190 ms
    // for the load stream alone).  Which chunk a workgroup takes changes no sum: partials are per (step, tile).
    const int t0 = 1 + ((int)gridDim.y - 1 - (int)blockIdx.y) * a.tchunk;
    const int t1 = min(t0 + a.tchunk, a.N);
    if (t0 >= t1) return;
    const double K = a.K, invK = a.invK;
    const int is_put = PUT < 0 ? a.is_put : PUT;
    // The chunk's discount factors go through the wave's LDS patch: a vector-memory load of
    // D[N-t] inside the loop would sit behind the row prefetch in the in-order vmcnt queue
    // and expose the prefetch latency every step.
    for (int i = lane; i < t1 - t0; i += 64) shD[wave][i] = a.D[a.N - (t0 + i)];
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
            // the VEC paths of a tile advance together, stage by stage: VEC independent dependency chains
            // (convert -> u -> select -> u^2 -> sums) instead of one, and consecutive accumulations go to
            // different accumulators -- the float64 pipe is 4 cycles deep per issue and a lone chain stalls it
            // Masking by multiplication: m = 1.0 in the money, 0.0 otherwise.  u*m and p*m are exact, so the
            // sums are those of the selected values bit for bit (a -0.0 added to a sum changes nothing), and
            // one 32-bit select + one multiply replace the four selects of (u, p) -- on this chip a
            // v_cndmask with an SGPR-pair mask costs as much as a float64 multiply (profiles/r02c_ubench.txt).
            // p needs no masking of its own except in its plain sum: u*m already zeroes u p and u^2 p.
            double u[VEC], m[VEC], u2[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) u[v] = fma((double)buf[k][v], invK, -1.0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float sf = buf[k][v];
                const bool itm = IS_PUT ? sf < thrk[k] : sf > thrk[k];
                cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(itm));
                m[v] = itm ? 1.0 : 0.0;
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) u[v] *= m[v];
#pragma unroll
            for (int v = 0; v < VEC; ++v) u2[v] = u[v] * u[v];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[1] += u[v];
                acc[2] += u2[v];
                acc[3] = fma(u2[v], u[v], acc[3]);
                acc[4] = fma(u2[v], u2[v], acc[4]);
                acc[5] = fma(pN[k][v], m[v], acc[5]);
                acc[6] = fma(u[v], pN[k][v], acc[6]);
                acc[7] = fma(u2[v], pN[k][v], acc[7]);
            }
        }
        const double d = shD[wave][t - t0];
        acc[0] = lane == 0 ? (double)cnt : 0.0;
        acc[5] *= d;
        acc[6] *= d;
        acc[7] *= d;
        const double s = wave_reduce8(acc, wl[wave]);
        // one 64-byte record per (step, tile): the eight result lanes write consecutive doubles, a workgroup's four
        // waves two whole cache lines -- no line of part1 is shared between workgroups (which run on different XCDs,
        // i.e. behind different L2s: with the quantity-major layout [t][q][tile] every line was written back piecemeal
        // from four of them: pass 1 of C3's shard 1.53-1.65 -> 1.48-1.50 ms, profiles/r04_xcd_mapping.txt)
        if ((lane & 7) == 0) a.part1[((size_t)t * a.ntiles + tg) * 8 + (lane >> 3)] = s;
    };
    // Rows are read with the nontemporal hint (each byte is used once per launch): 253 -> 218 us at
    // C2 (pass 2 reads its rows the same way; on the generator's stores the hint only moves time between kernels).
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

// ------------------------------------------------------------------ antithetic-folded storage (GBM)
// For GBM the antithetic partner of a path is a FUNCTION of the path: S_t = S0 exp(sum(a + b z_i)) and
// S'_t = S0 exp(sum(a - b z_i)) give S_t S'_t = S0^2 exp(2 a t) =: C_t, a per-step constant.  The fused pricing therefore
// stores only the first partner of every pair -- half the matrix: half the bytes written by the generator and read by both
// sweeps -- and the sweeps price both partners from every spot they load.  The partner enters all arithmetic through its
// moneyness, in float64 and never rounded to float32:
//      u' = x' - 1 = (C_t / K) / S_t - 1 = fma(cK_t, rcp(S_t), -1),
// its payoff is -K u' (put) / K u' (call), in the money <=> that is > 0, its continuation value the fit at u'.  These three
// definitions are shared by pass 1, pass 2 and the valuation (fold_u / fold_pay), so the sweeps agree bit for bit on which
// partner rows exist and what they are worth; the CPU restatement the tests check against repeats them (orc_lsm_two_pass_folded).
// (1 / S_t: v_rcp_f64 is good to 2^-23; ONE Newton step takes it to 2^-46 = 1.4e-14 relative -- the partner's moneyness is
//  then 1.4e-14 (absolute) from the exactly divided one, nine orders below the float32 rounding of the spot it is made from)
__device__ __forceinline__ double fold_u(double cK_t, float s)
{
    const double x = (double)s;
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(cK_t, r, -1.0);
}
__device__ __forceinline__ double fold_pay(double u, double K, int is_put) { return is_put ? -K * u : K * u; }
constexpr int kFoldMaxChunk = 64;  // steps per workgroup of the folded pass 1, at most

// Pass 1 on the folded matrix: as lsm_pass1_body, every loaded spot contributing its own row and its partner's.
// a.M = stored columns (pairs); TPW tiles of 64 * VEC columns = 2 * TPW * 64 * VEC paths per wave and step.
template <int VEC, int TPW, int PUT = -1>
__device__ __forceinline__ void lsm_pass1_fold_body(Pass1Args a)
{
    __shared__ double wl[kBlock / 64][kWaveRedDoubles];
    __shared__ double shD[kBlock / 64][kFoldMaxChunk];  // (short tables: LDS must not cap the waves per SIMD)
    __shared__ double shC[kBlock / 64][kFoldMaxChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tg = (int64_t)xcd_block((int)blockIdx.x, (int)gridDim.x) * (kBlock / 64) + wave;
    if (tg >= a.ntiles) return;  // whole wave leaves; no workgroup barrier below
    const int64_t base = tg * (64 * VEC * TPW) + (int64_t)lane * VEC;
    const int t0 = 1 + ((int)gridDim.y - 1 - (int)blockIdx.y) * a.tchunk;
    const int t1 = min(t0 + a.tchunk, a.N);
    if (t0 >= t1) return;
    const double K = a.K, invK = a.invK;
    const int is_put = PUT < 0 ? a.is_put : PUT;
    for (int i = lane; i < t1 - t0; i += 64) {
        shD[wave][i] = a.D[a.N - (t0 + i)];
        shC[wave][i] = a.cK[t0 + i];
    }
    const float* colp[TPW];
    double pNA[TPW][VEC], pNB[TPW][VEC];
    bool valid[TPW];
    const double cKN = a.cK[a.N];
    float sn[TPW][VEC];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int64_t j = base + (int64_t)k * 64 * VEC;
        valid[k] = j < a.M;
        colp[k] = a.S + (valid[k] ? j : 0);
        loadf<VEC>(colp[k] + (int64_t)a.N * a.ld, sn[k]);
    }
    auto load_rows = [&](float (&buf)[TPW][VEC], int t) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) loadf_stream<VEC>(colp[k] + (int64_t)t * a.ld, buf[k]);
    };
    // the chunk's first two rows are requested BEFORE the terminal row is worked on (a short chunk lives ~11 us: a second
    // memory round trip in its prologue would be a fifth of that)
    float bufA[TPW][VEC], bufB[TPW][VEC], bufC[TPW][VEC];
    const int tl = t1 - 1;
    load_rows(bufA, t0);
    load_rows(bufB, min(t0 + 1, tl));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const double pa = payoff_d(sn[k][v], K, is_put), pb = fold_pay(fold_u(cKN, sn[k][v]), K, is_put);
            pNA[k][v] = (valid[k] && pa > 0.0) ? pa : 0.0;
            pNB[k][v] = (valid[k] && pb > 0.0) ? pb : 0.0;
        }
    }
    const float thr = itm_threshold(K, is_put);
    float thrk[TPW];  // padding tiles: a threshold no price passes
#pragma unroll
    for (int k = 0; k < TPW; ++k) thrk[k] = valid[k] ? thr : (is_put ? -__builtin_inff() : __builtin_inff());
    unsigned long long vmask[TPW];  // lanes whose columns of tile k exist
#pragma unroll
    for (int k = 0; k < TPW; ++k) vmask[k] = __builtin_amdgcn_ballot_w64(valid[k]);
    auto process = [&](auto put_tag, const float (&buf)[TPW][VEC], int t) {
        constexpr bool IS_PUT = decltype(put_tag)::value;
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.0;
        int cnt = 0;
        const double ck = shC[wave][t - t0];
        // One row per spot in the common case: of the two partners of a pair at most one is in the money unless the spot lies
        // between K and C_t / K (S0 = K: an empty or narrow band), so the sums take the PRIMARY row -- the stored path's if it
        // is in the money, else the partner's -- and the partner's row a second time only where some lane of the wave has
        // both in the money (a wave-uniform branch per spot slot; deep in the money it is always taken).
        auto add_row = [&](double u, double y, double m) {  // u, y already zero where m is
            const double u2 = u * u;
            acc[1] += u;
            acc[2] += u2;
            acc[3] = fma(u2, u, acc[3]);
            acc[4] = fma(u2, u2, acc[4]);
            acc[5] = fma(y, m, acc[5]);
            acc[6] = fma(u, y, acc[6]);
            acc[7] = fma(u2, y, acc[7]);
        };
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float sf = buf[k][v];
                const double ua = fma((double)sf, invK, -1.0);
                const double ub = fold_u(ck, sf);
                const bool ia = IS_PUT ? sf < thrk[k] : sf > thrk[k];
                // the partner: in the money <=> its payoff -K u' (put) / K u' (call) > 0; padding columns never are
                const bool ibc = IS_PUT ? ub < 0.0 : ub > 0.0;
                // (the wave's lane masks straight from the two compares and combined in scalar registers: a ballot of
                //  `ia || ib` makes the compiler materialise the boolean in a vector register and compare it again)
                const unsigned long long ma = __builtin_amdgcn_ballot_w64(ia);
                const unsigned long long mb = __builtin_amdgcn_ballot_w64(ibc) & vmask[k];
                const bool ib = valid[k] && ibc;
                cnt += __builtin_popcountll(ma | mb);  // rows of this spot: any + both
                const double mp = (ia || ib) ? 1.0 : 0.0;
                add_row((ia ? ua : ub) * mp, ia ? pNA[k][v] : pNB[k][v], mp);
                const unsigned long long bb = ma & mb;
                if (bb != 0) {
                    asm volatile("; a lane with both partners in the money" ::);  // (keeps the branch: no if-conversion)
                    cnt += __builtin_popcountll(bb);
                    const bool both = ia && ib;
                    const double ms = both ? 1.0 : 0.0;
                    add_row(ub * ms, pNB[k][v], ms);
                }
            }
        }
        const double d = shD[wave][t - t0];
        acc[0] = lane == 0 ? (double)cnt : 0.0;
        acc[5] *= d;
        acc[6] *= d;
        acc[7] *= d;
        const double s = wave_reduce8(acc, wl[wave]);
        if ((lane & 7) == 0) a.part1[((size_t)t * a.ntiles + tg) * 8 + (lane >> 3)] = s;
    };
    auto sweep = [&](auto put_tag) {
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
    // One workgroup per step: a thread adds the 64-byte records of tiles tid, tid + 256, ... (all eight quantities of
    // a tile in one contiguous read), then every quantity goes through the same tree -- xor-shuffles inside a wave,
    // waves paired (0 + 1) + (2 + 3).  Per quantity that is the order of the former one-workgroup-per-(step,
    // quantity) kernel over the [t][q][tile] layout: same bits.
    __shared__ double sh[kBlock / 64][8];
    const int tid = threadIdx.x;
    const int t = blockIdx.x + 1;
    const double2* pp = reinterpret_cast<const double2*>(part1 + (size_t)t * ntiles * 8);
    double s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = 0.0;
    for (int64_t i = tid; i < ntiles; i += kBlock) {
        double2 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = pp[i * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[2 * k] += r[k].x;
            s[2 * k + 1] += r[k].y;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) sh[tid >> 6][q] = s[q];
    }
    __syncthreads();
    if (tid < 8) gmom[(size_t)t * 8 + tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
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
    // non-null: the fits are not given but solved here from the reduced moment table [N+1][8] (every workgroup
    // solves all steps itself, one step per thread: cheaper than a dependent launch in between), and workgroup 0
    // writes them to betas_out [N+1][4]
    const double* gmom = nullptr;
    double* betas_out = nullptr;
    const double* cK = nullptr;  // antithetic-folded storage (lsm_pass2_fold_body): see Pass1Args::cK; M = stored columns
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
    if (a.gmom) {
        for (int t = tid; t <= N; t += kBlock) {
            double m[8], beta[3] = {0.0, 0.0, 0.0};
            const bool inner = t >= 1 && t < N;
#pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = inner ? a.gmom[(size_t)t * 8 + q] : 0.0;
            if (inner) solve_poly2(m, beta);  // the very function lsm_solve_all_kernel runs: same bits
            const bool fit = inner && m[0] > 0.5;
            sh_b[4 * t] = fit ? beta[0] : __builtin_huge_val();
            sh_b[4 * t + 1] = fit ? beta[1] : 0.0;
            sh_b[4 * t + 2] = fit ? beta[2] : 0.0;
            sh_b[4 * t + 3] = 0.0;
            if (blockIdx.x == 0 && inner && a.betas_out) {
                double* bo = a.betas_out + (size_t)t * 4;
                bo[0] = beta[0]; bo[1] = beta[1]; bo[2] = beta[2]; bo[3] = m[0];
            }
        }
    } else {
        for (int k = tid; k < (N + 1) * 4; k += kBlock) {
            const int t = k >> 2;
            const bool fit = t >= 1 && t < N && a.betas[(size_t)t * 4 + 3] > 0.5;
            sh_b[k] = fit ? a.betas[k] : ((k & 3) == 0 ? __builtin_huge_val() : 0.0);
        }
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
        // full blocks of U rows: U unconditional 16-byte loads in flight per lane (4 and 16 measured slower).
        // Rows are read with the nontemporal hint, each byte being used once per launch: 0.187 -> 0.169 ms at C2
        // (tools/exp_pass2.sh; in round 1, before the sweep was restructured, the same hint did nothing here).
        constexpr int U = 8;
        int t = N - 1;
        const float* col = a.S + j;
        for (; t >= U && live(); t -= U) {
            float st[U][VEC];
#pragma unroll
            for (int k = 0; k < U; ++k) loadf_stream<VEC>(col + (int64_t)(t - k) * a.ld, st[k]);
#pragma unroll
            for (int k = 0; k < U; ++k) decide(st[k], t - k);
        }
        for (; t >= 1 && live(); --t) {
            float st[VEC];
            loadf_stream<VEC>(col + (int64_t)t * a.ld, st);
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

// Pass 2 on the folded matrix: a thread walks its VEC stored columns backward and decides for both partners of each from
// every spot it loads; it stops when all 2 VEC paths have exercised.  A partner that exercises remembers the STORED
// path's spot at that step (its own value is fold_pay(fold_u(cK[t], spot))): the valuation below recomputes it with the
// very expressions of the decision.  The fits' table keeps cK[t] in its fourth slot.
template <int VEC, int PUT = -1>
__device__ __forceinline__ void lsm_pass2_fold_body(Pass2Args a)
{
    if ((int)blockIdx.x >= a.nblk) return;
    __shared__ double red[kNQ * kRedStride];
    extern __shared__ double sh_b[];  // [N+1][4]: b0, b1, b2, cK
    const int tid = threadIdx.x;
    const int N = a.N;
    if (a.gmom) {
        for (int t = tid; t <= N; t += kBlock) {
            double m[8], beta[3] = {0.0, 0.0, 0.0};
            const bool inner = t >= 1 && t < N;
#pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = inner ? a.gmom[(size_t)t * 8 + q] : 0.0;
            if (inner) solve_poly2(m, beta);
            const bool fit = inner && m[0] > 0.5;
            sh_b[4 * t] = fit ? beta[0] : __builtin_huge_val();
            sh_b[4 * t + 1] = fit ? beta[1] : 0.0;
            sh_b[4 * t + 2] = fit ? beta[2] : 0.0;
            sh_b[4 * t + 3] = a.cK[t];
            if (blockIdx.x == 0 && inner && a.betas_out) {
                double* bo = a.betas_out + (size_t)t * 4;
                bo[0] = beta[0]; bo[1] = beta[1]; bo[2] = beta[2]; bo[3] = m[0];
            }
        }
    } else {
        for (int k = tid; k < (N + 1) * 4; k += kBlock) {
            const int t = k >> 2;
            const bool fit = t >= 1 && t < N && a.betas[(size_t)t * 4 + 3] > 0.5;
            sh_b[k] = (k & 3) == 3 ? a.cK[t] : (fit ? a.betas[k] : ((k & 3) == 0 ? __builtin_huge_val() : 0.0));
        }
    }
    __syncthreads();
    const double K = a.K, invK = a.invK;
    const int is_put = PUT < 0 ? a.is_put : PUT;  // (a compile-time side: the payoff is one subtraction, not two and a select)
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)a.nblk * kBlock * VEC;
    for (int64_t j = ((int64_t)blockIdx.x * kBlock + tid) * VEC; j < a.M; j += stride) {
        float sxa[VEC], sxb[VEC];
        int32_t texa[VEC], texb[VEC];
        loadf<VEC>(a.S + (int64_t)N * a.ld + j, sxa);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            sxb[v] = sxa[v];
            texa[v] = texb[v] = N;
        }
        auto decide = [&](const float (&row)[VEC], int t) {
            const double b0 = sh_b[4 * t], b1 = sh_b[4 * t + 1], b2 = sh_b[4 * t + 2], ck = sh_b[4 * t + 3];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double sd = (double)row[v];
                const double imm = is_put ? K - sd : sd - K;
                const double u = fma(sd, invK, -1.0);
                const double cont = fma(u, fma(u, b2, b1), b0);
                const bool ex = (texa[v] == N) & (imm > 0.0) & (imm > cont);
                sxa[v] = ex ? row[v] : sxa[v];
                texa[v] = ex ? t : texa[v];
                const double ub = fold_u(ck, row[v]);
                const double immb = fold_pay(ub, K, is_put);
                const double contb = fma(ub, fma(ub, b2, b1), b0);
                const bool exb = (texb[v] == N) & (immb > 0.0) & (immb > contb);
                sxb[v] = exb ? row[v] : sxb[v];
                texb[v] = exb ? t : texb[v];
            }
        };
        auto live = [&]() {
            bool l = false;
#pragma unroll
            for (int v = 0; v < VEC; ++v) l |= (texa[v] == N) | (texb[v] == N);
            return l;
        };
        constexpr int U = 8;
        int t = N - 1;
        const float* col = a.S + j;
        // U rows per batch, the NEXT batch requested before the current one is worked on (the decisions of a batch are a
        // dependent chain of ~450 vector instructions: without the look-ahead every batch waited out a whole memory round
        // trip with 4 waves per SIMD to cover it).  Rows below 1 are clamped to row 1 and not decided.
        auto fetch = [&](float (&b)[U][VEC], int tt) {
#pragma unroll
            for (int k = 0; k < U; ++k) loadf_stream<VEC>(col + (int64_t)max(tt - k, 1) * a.ld, b[k]);
        };
        auto work = [&](const float (&b)[U][VEC], int tt) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (tt - k >= 1) decide(b[k], tt - k);
        };
        float bA[U][VEC], bB[U][VEC];
        if (t >= 1) fetch(bA, t);
        while (t >= 1 && live()) {
            if (t - U >= 1) fetch(bB, t - U);
            __builtin_amdgcn_sched_barrier(0);
            work(bA, t);
            t -= U;
            if (!(t >= 1 && live())) break;
            if (t - U >= 1) fetch(bA, t - U);
            __builtin_amdgcn_sched_barrier(0);
            work(bB, t);
            t -= U;
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            double p = payoff_d(sxa[v], K, is_put);
            p = p > 0.0 ? p : 0.0;
            const double cf = p * a.D[texa[v] - 1];
            double pb = fold_pay(fold_u(sh_b[4 * texb[v] + 3], sxb[v]), K, is_put);
            pb = pb > 0.0 ? pb : 0.0;
            const double cfb = pb * a.D[texb[v] - 1];
            acc[0] += cf;
            acc[1] += cf * cf;
            acc[2] += (texa[v] < N) ? 1.0 : 0.0;
            acc[3] += (cf == 0.0) ? 1.0 : 0.0;
            acc[0] += cfb;
            acc[1] += cfb * cfb;
            acc[2] += (texb[v] < N) ? 1.0 : 0.0;
            acc[3] += (cfb == 0.0) ? 1.0 : 0.0;
        }
    }
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) a.part[(size_t)(tid >> 3) * a.pstride + blockIdx.x] = s;
}

// ------------------------------------------------------------------ valuation + finalize
struct FinalArgs {
    float* sx;
    int32_t* tex;
    const float* live;  // per-step reference flow: >= 0 = never exercised, the value is S_N -> (S_N, N); else null
    int64_t M;
    int N, is_put, tval, fill_state;  // fill_state: write (S_N, N) into sx / tex of unexercised paths
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
        if (a.live) {
            float sn[VEC];
            loadf<VEC>(a.live + j, sn);
            bool any = false;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (!(sn[v] < 0.0f)) { sx[v] = sn[v]; tex[v] = a.N; any = true; }
            }
            if (a.fill_state && any) {
                storef<VEC>(a.sx + j, sx);
                storei<VEC>(a.tex + j, tex);
            }
        }
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
                                                  int pstride, int gstride = 8)
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
    for (int t = 1 + tid; t < N; t += kBlock) acc[4] += gmom[(size_t)t * gstride];
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) result[tid >> 3] = s;
}

}  // namespace omc
