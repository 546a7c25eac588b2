// omc_paths_dev.h -- device code of the path generators (gfx950), shared by the single-problem
// launchers (omc_paths.hip) and the batched launchers (omc_batch.hip).  Bodies only look at
// blockIdx.x; the batch dimension is resolved by the __global__ wrapper.
#pragma once
#include "omc_device.h"
#include "omc_kernels.h"

namespace omc {


template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC])
{
    // plain stores: the nontemporal hint on these rows only moves time from this kernel to the next reader
    // (measured)
    if constexpr (VEC == 1) *p = v[0];
    else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    else *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------ GBM
// ANTI: pair p -> columns p and p + P (partner of column j is j + M/2, as the reference
// lays it out).  !ANTI: P independent paths, one column each.
struct HestonC {
    float dtf, kdt, theta, xi, rho, rho2, rdt_l2, hdt_l2, l2e;
    float sqdt, rdt, xi_sqdt;  // sqrt(dt), r*dt, xi*sqrt(dt)
    float l2e_sqdt;            // log2(e)*sqrt(dt)
};

// one argument block for both generators (Heston ignores a/b, GBM ignores v_init/hc)
struct PathArgs {
    float* S;
    int64_t ld, P;
    int n_steps;
    float s_init, a, b, v_init;
    HestonC hc;
    uint32_t k0, k1, stream;
    uint64_t pair_offset;
};

template <int VEC, bool ANTI>
__device__ __forceinline__ void gbm_paths_body(const PathArgs& g)
{
    float* __restrict__ S = g.S;
    const int64_t ld = g.ld, P = g.P;
    const int n_steps = g.n_steps;
    const float s_init = g.s_init, a = g.a, b = g.b;
    const uint32_t k0 = g.k0, k1 = g.k1, stream = g.stream;
    const uint64_t pair_offset = g.pair_offset;
    const int64_t p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
    if (p0 >= P) return;
    float s[VEC], sa[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = sa[v] = s_init;
    float* row = S + p0;
    store_vec<VEC>(row, s);
    if (ANTI) store_vec<VEC>(row + P, sa);

    const int nblk = (n_steps + 3) >> 2;
    int t = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        float z[VEC][4];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            normals4(pair_offset + (uint64_t)(p0 + v), (uint32_t)blk, stream, k0, k1, z[v]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (++t > n_steps) break;
            row += ld;
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[v] = s[v] * fast_exp2(__builtin_fmaf(b, z[v][i], a));
            store_vec<VEC>(row, s);
            if (ANTI) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    sa[v] = sa[v] * fast_exp2(__builtin_fmaf(-b, z[v][i], a));
                store_vec<VEC>(row + P, sa);
            }
        }
    }
}

// ------------------------------------------------------------------ Heston
// One Euler step of an ANTITHETIC PAIR (normals (z1, z2) and (-z1, -z2)); the operation order is part
// of the numerics contract that the test-side CPU restatement follows as well.  What the
// two partners share is computed once: w2 = rho z1 + rho2 z2, tw = xi sqrt(dt) w2, az = log2(e) sqrt(dt) z1
// -- the partner's are their exact negations -- so a path costs max, sqrt, sub, 4 fma, exp2, mul (, max).
// max(x, lo) as ONE v_max_f32: fmaxf() costs two instructions here (hipcc first quiets a possible
// signalling NaN with v_max x, x, x; it also folds v_med3(x, lo, inf) back into that pair)
__device__ __forceinline__ float floor_at(float x, float lo)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo));
    return r;
}

template <int SCHEME>
__device__ __forceinline__ void heston_path_step(const HestonC& c, float tw, float az, float& s, float& v)
{
    const float vp = floor_at(v, 0.0f);
    const float sq = __builtin_amdgcn_sqrtf(vp);
    const float base = SCHEME ? v : vp;
    const float vn = __builtin_fmaf(sq, tw, __builtin_fmaf(c.kdt, c.theta - vp, base));
    const float arg = __builtin_fmaf(sq, az, __builtin_fmaf(-c.hdt_l2, vp, c.rdt_l2));
    s = s * fast_exp2(arg);
    v = SCHEME ? vn : floor_at(vn, 0.0f);
}

template <int SCHEME>
__device__ __forceinline__ void heston_pair_step(const HestonC& c, float z1, float z2, float& s, float& v,
                                                 float& sa, float& vb)
{
    if constexpr (SCHEME == 2) {
        // The calibrator's own scheme (heston_calibration.py:242-255): variance floored at 1e-8
        // before use and at store, ARITHMETIC Euler for S (S may cross zero; so be it).
        const float w2 = __builtin_fmaf(c.rho, z1, c.rho2 * z2);
        {
            const float vp = floor_at(v, 1e-8f);
            const float sq = __builtin_amdgcn_sqrtf(vp);
            const float vn = __builtin_fmaf(c.xi_sqdt * sq, w2, __builtin_fmaf(c.kdt, c.theta - vp, vp));
            s = __builtin_fmaf(s, __builtin_fmaf(sq * c.sqdt, z1, c.rdt), s);
            v = floor_at(vn, 1e-8f);
        }
        {
            const float vp = floor_at(vb, 1e-8f);
            const float sq = __builtin_amdgcn_sqrtf(vp);
            const float vn = __builtin_fmaf(c.xi_sqdt * sq, -w2, __builtin_fmaf(c.kdt, c.theta - vp, vp));
            sa = __builtin_fmaf(sa, __builtin_fmaf(sq * c.sqdt, -z1, c.rdt), sa);
            vb = floor_at(vn, 1e-8f);
        }
        return;
    } else {
        const float w2 = __builtin_fmaf(c.rho, z1, c.rho2 * z2);
        const float tw = c.xi_sqdt * w2;
        const float az = c.l2e_sqdt * z1;
        heston_path_step<SCHEME>(c, tw, az, s, v);
        heston_path_step<SCHEME>(c, -tw, -az, sa, vb);
    }
}

// One Philox block per pair per TWO steps: words (0,1) -> (z1,z2) of the odd step,
// words (2,3) -> (z1,z2) of the even step.  The variance never leaves registers.
template <int VEC, int SCHEME>
__device__ __forceinline__ void heston_paths_body(const PathArgs& g)
{
    float* __restrict__ S = g.S;
    const int64_t ld = g.ld, P = g.P;
    const int n_steps = g.n_steps;
    const float s_init = g.s_init, v_init = g.v_init;
    const HestonC c = g.hc;
    const uint32_t k0 = g.k0, k1 = g.k1, stream = g.stream;
    const uint64_t pair_offset = g.pair_offset;
    const int64_t p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
    if (p0 >= P) return;
    float s[VEC], sa[VEC], va[VEC], vb[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        s[v] = sa[v] = s_init;
        va[v] = vb[v] = v_init;
    }
    float* row = S + p0;
    store_vec<VEC>(row, s);
    store_vec<VEC>(row + P, sa);
    const int nblk = (n_steps + 1) >> 1;
    int t = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        float z[VEC][4];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            normals4(pair_offset + (uint64_t)(p0 + v), (uint32_t)blk, stream, k0, k1, z[v]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (++t > n_steps) break;
            row += ld;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                heston_pair_step<SCHEME>(c, z[v][2 * i], z[v][2 * i + 1], s[v], va[v], sa[v], vb[v]);
            store_vec<VEC>(row, s);
            store_vec<VEC>(row + P, sa);
        }
    }
}

// ------------------------------------------------------------------ terminal-only (European)
// Replaces price_european_streaming (options_model_3.py:382-437): the reference builds the
// whole [N+1][chunk] matrix per 500-path chunk to read its last row; here S lives in
// registers and only block partial sums {sum, sumsq, n_zero} of the discounted payoff leave.
struct TermArgs {
    int64_t P;
    int n_steps, is_put;
    float s_init, a, b, v_init;
    HestonC hc;
    uint32_t k0, k1, stream;
    uint64_t pair_offset;
    double K, df;
    double* part;  // [8][pstride]
    int nblk, pstride;
};

__device__ __forceinline__ void add_payoff(double (&acc)[8], float s, double K, int is_put, double df)
{
    double p = payoff_d(s, K, is_put);
    p = p > 0.0 ? p * df : 0.0;
    acc[0] += p;
    acc[1] += p * p;
    acc[3] += (p == 0.0) ? 1.0 : 0.0;
}

// MODEL 0 GBM (ANTI selectable), MODEL 1/2/3 Heston scheme 0/1/2 (always antithetic)
template <int MODEL, bool ANTI>
__device__ __forceinline__ void terminal_body(const TermArgs& a)
{
    if ((int)blockIdx.x >= a.nblk) return;
    __shared__ double red[kNQ * kRedStride];
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)a.nblk * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < a.P; p += stride) {
        float s = a.s_init, sa = a.s_init, va = a.v_init, vb = a.v_init, z[4];
        if (MODEL == 0) {
            for (int t = 0; t < a.n_steps; ++t) {
                if ((t & 3) == 0)
                    normals4(a.pair_offset + (uint64_t)p, (uint32_t)(t >> 2), a.stream, a.k0, a.k1, z);
                s = s * fast_exp2(__builtin_fmaf(a.b, z[t & 3], a.a));
                if (ANTI) sa = sa * fast_exp2(__builtin_fmaf(-a.b, z[t & 3], a.a));
            }
        } else {
            for (int t = 0; t < a.n_steps; ++t) {
                const int i = t & 1;
                if (i == 0)
                    normals4(a.pair_offset + (uint64_t)p, (uint32_t)(t >> 1), a.stream, a.k0, a.k1, z);
                heston_pair_step<MODEL - 1>(a.hc, z[2 * i], z[2 * i + 1], s, va, sa, vb);
            }
        }
        add_payoff(acc, s, a.K, a.is_put, a.df);
        if (ANTI) add_payoff(acc, sa, a.K, a.is_put, a.df);
    }
    const double r = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0)
        a.part[(size_t)(threadIdx.x >> 3) * a.pstride + blockIdx.x] = r;
}

}  // namespace omc
