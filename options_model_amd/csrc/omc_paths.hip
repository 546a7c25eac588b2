// omc_paths.hip -- path generation kernels for gfx950 (MI355X).
//
// Replaces the reference's per-step numpy/torch op chains
//   GBM    options_model_3/options_model_3.py:473-480   (GPU intent: option_model_3_gpu.py:117-185)
//   Heston options_model_3/options_model_3.py:211-251   (GPU intent: option_model_3_gpu.py:187-248)
// with ONE launch each: a thread owns VEC antithetic pairs, keeps S (and the Heston
// variance) in registers for the whole time loop, draws its normals from a counter RNG
// (no state, no reads) and streams the [step][path] matrix out with 4*VEC-byte stores.
// The kernel is pure HBM write traffic: (n_steps+1) * n_paths * 4 bytes.
#include "omc_device.h"
#include "omc_kernels.h"

namespace omc {

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC])
{
    if constexpr (VEC == 1) {
        *p = v[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ------------------------------------------------------------------ GBM
// ANTI: pair p -> columns p and p + P (partner of column j is j + M/2, as the reference
// lays it out).  !ANTI: P independent paths, one column each.
template <int VEC, bool ANTI>
__global__ __launch_bounds__(kBlock) void gbm_paths_kernel(float* __restrict__ S, int64_t ld,
                                                           int64_t P, int n_steps, float s_init,
                                                           float a, float b, uint32_t k0,
                                                           uint32_t k1, uint32_t stream,
                                                           uint64_t pair_offset)
{
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
struct HestonC {
    float dtf, kdt, theta, xi, rho, rho2, rdt_l2, hdt_l2, l2e;
};

// one Euler step; the operation order is part of the numerics contract that the
// test-side CPU restatement follows as well
template <int SCHEME>
__device__ __forceinline__ void heston_step(const HestonC& c, float z1, float z2, float& s, float& v)
{
    const float vp = fmaxf(v, 0.0f);
    const float sq = __builtin_amdgcn_sqrtf(vp * c.dtf);
    const float w2 = __builtin_fmaf(c.rho, z1, c.rho2 * z2);
    const float base = SCHEME ? v : vp;
    const float vn = __builtin_fmaf(c.xi * sq, w2, __builtin_fmaf(c.kdt, c.theta - vp, base));
    const float arg = __builtin_fmaf(sq * c.l2e, z1, __builtin_fmaf(-c.hdt_l2, vp, c.rdt_l2));
    s = s * fast_exp2(arg);
    v = SCHEME ? vn : fmaxf(vn, 0.0f);
}

// One Philox block per pair per TWO steps: words (0,1) -> (z1,z2) of the odd step,
// words (2,3) -> (z1,z2) of the even step.  The variance never leaves registers.
template <int VEC, int SCHEME>
__global__ __launch_bounds__(kBlock) void heston_paths_kernel(float* __restrict__ S, int64_t ld,
                                                              int64_t P, int n_steps, float s_init,
                                                              float v_init, HestonC c, uint32_t k0,
                                                              uint32_t k1, uint32_t stream,
                                                              uint64_t pair_offset)
{
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
            for (int v = 0; v < VEC; ++v) {
                heston_step<SCHEME>(c, z[v][2 * i], z[v][2 * i + 1], s[v], va[v]);
                heston_step<SCHEME>(c, -z[v][2 * i], -z[v][2 * i + 1], sa[v], vb[v]);
            }
            store_vec<VEC>(row, s);
            store_vec<VEC>(row + P, sa);
        }
    }
}

// ------------------------------------------------------------------ injected normals
// Parity mode: the normals come from HBM (Zhalf [n_steps][ldz], row t-1 drives step t,
// options_model_3.py:475-480) instead of Philox.  Same arithmetic as the kernels above.
template <bool ANTI>
__global__ __launch_bounds__(kBlock) void gbm_from_normals_kernel(float* __restrict__ S, int64_t ld,
                                                                  int64_t P, int n_steps,
                                                                  float s_init, float a, float b,
                                                                  const float* __restrict__ Z,
                                                                  int64_t ldz)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float s = s_init, sa = s_init;
    S[p] = s;
    if (ANTI) S[p + P] = sa;
    for (int t = 1; t <= n_steps; ++t) {
        const float z = Z[(int64_t)(t - 1) * ldz + p];
        s = s * fast_exp2(__builtin_fmaf(b, z, a));
        S[(int64_t)t * ld + p] = s;
        if (ANTI) {
            sa = sa * fast_exp2(__builtin_fmaf(-b, z, a));
            S[(int64_t)t * ld + p + P] = sa;
        }
    }
}

template <int SCHEME>
__global__ __launch_bounds__(kBlock) void heston_from_normals_kernel(
    float* __restrict__ S, int64_t ld, int64_t P, int n_steps, float s_init, float v_init,
    HestonC c, const float* __restrict__ Z1, const float* __restrict__ Z2, int64_t ldz)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float s = s_init, sa = s_init, va = v_init, vb = v_init;
    S[p] = s;
    S[p + P] = sa;
    for (int t = 1; t <= n_steps; ++t) {
        const float z1 = Z1[(int64_t)(t - 1) * ldz + p], z2 = Z2[(int64_t)(t - 1) * ldz + p];
        heston_step<SCHEME>(c, z1, z2, s, va);
        heston_step<SCHEME>(c, -z1, -z2, sa, vb);
        S[(int64_t)t * ld + p] = s;
        S[(int64_t)t * ld + p + P] = sa;
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
    double* part;  // [8][kMaxLsmBlocks]
};

__device__ __forceinline__ void add_payoff(double (&acc)[8], float s, double K, int is_put, double df)
{
    double p = payoff_d(s, K, is_put);
    p = p > 0.0 ? p * df : 0.0;
    acc[0] += p;
    acc[1] += p * p;
    acc[3] += (p == 0.0) ? 1.0 : 0.0;
}

// MODEL 0 GBM (ANTI selectable), MODEL 1/2 Heston scheme 0/1 (always antithetic)
template <int MODEL, bool ANTI>
__global__ __launch_bounds__(kBlock) void terminal_kernel(TermArgs a)
{
    __shared__ double red[kNQ * kRedStride];
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
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
                heston_step<MODEL - 1>(a.hc, z[2 * i], z[2 * i + 1], s, va);
                heston_step<MODEL - 1>(a.hc, -z[2 * i], -z[2 * i + 1], sa, vb);
            }
        }
        add_payoff(acc, s, a.K, a.is_put, a.df);
        if (ANTI) add_payoff(acc, sa, a.K, a.is_put, a.df);
    }
    const double r = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0)
        a.part[(size_t)(threadIdx.x >> 3) * kMaxLsmBlocks + blockIdx.x] = r;
}

// ------------------------------------------------------------------ RNG test taps
__global__ void philox_kat_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* c = in + 6 * i;
    const U4 o = philox4x32_10(c[0], c[1], c[2], c[3], c[4], c[5]);
    out[4 * i + 0] = o.x;
    out[4 * i + 1] = o.y;
    out[4 * i + 2] = o.z;
    out[4 * i + 3] = o.w;
}

__global__ __launch_bounds__(kBlock) void gbm_normals_kernel(float* __restrict__ Z, int64_t ldz,
                                                             int64_t P, int n_steps, uint32_t k0,
                                                             uint32_t k1, uint32_t stream,
                                                             uint64_t pair_offset)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float z[4];
    for (int t = 0; t < n_steps; ++t) {
        if ((t & 3) == 0) normals4(pair_offset + (uint64_t)p, (uint32_t)(t >> 2), stream, k0, k1, z);
        Z[(int64_t)t * ldz + p] = z[t & 3];
    }
}

// ------------------------------------------------------------------ host launchers
static inline HestonC make_heston(double r, double T, int n_steps, double kappa, double theta,
                                  double xi, double rho)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    HestonC c;
    c.dtf = (float)dt;
    c.kdt = (float)(kappa * dt);
    c.theta = (float)theta;
    c.xi = (float)xi;
    c.rho = (float)rho;
    c.rho2 = (float)sqrt(1.0 - rho * rho);
    c.rdt_l2 = (float)(r * dt * L2E);
    c.hdt_l2 = (float)(0.5 * dt * L2E);
    c.l2e = (float)L2E;
    return c;
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static inline unsigned grid_for(int64_t work_items)
{
    return (unsigned)((work_items + kBlock - 1) / kBlock);
}

hipError_t launch_gbm_paths(hipStream_t st, float* S, int64_t ld, int64_t n_paths, int n_steps,
                            double S0, double r, double sigma, double T, uint64_t seed,
                            uint32_t stream, uint64_t pair_offset, int antithetic, int vec_hint)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    const float a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    const float b = (float)(sigma * sqrt(dt) * L2E);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t P = antithetic ? n_paths / 2 : n_paths;
    if (P <= 0) return hipSuccess;
    int vec = vec_hint > 0 ? vec_hint : 4;
    // VEC-wide stores need every row start (and the antithetic half) aligned
    while (vec > 1 && !((P % vec) == 0 && (ld % vec) == 0 && ((uintptr_t)S % (4 * vec)) == 0)) vec >>= 1;
    const float s0 = (float)S0;
#define OMC_LAUNCH_GBM(V, A)                                                                       \
    hipLaunchKernelGGL((gbm_paths_kernel<V, A>), dim3(grid_for((P + V - 1) / V)), dim3(kBlock), 0, \
                       st, S, ld, P, n_steps, s0, a, b, k0, k1, stream, pair_offset)
    if (antithetic) {
        if (vec == 4) OMC_LAUNCH_GBM(4, true);
        else if (vec == 2) OMC_LAUNCH_GBM(2, true);
        else OMC_LAUNCH_GBM(1, true);
    } else {
        if (vec == 4) OMC_LAUNCH_GBM(4, false);
        else if (vec == 2) OMC_LAUNCH_GBM(2, false);
        else OMC_LAUNCH_GBM(1, false);
    }
#undef OMC_LAUNCH_GBM
    return hipGetLastError();
}

hipError_t launch_heston_paths(hipStream_t st, float* S, int64_t ld, int64_t n_paths, int n_steps,
                               double S0, double r, double T, double v0, double kappa,
                               double theta, double xi, double rho, uint64_t seed, uint32_t stream,
                               uint64_t pair_offset, int scheme, int vec_hint)
{
    const HestonC c = make_heston(r, T, n_steps, kappa, theta, xi, rho);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t P = n_paths / 2;
    if (P <= 0) return hipSuccess;
    int vec = vec_hint > 0 ? vec_hint : 4;
    while (vec > 1 && !((P % vec) == 0 && (ld % vec) == 0 && ((uintptr_t)S % (4 * vec)) == 0)) vec >>= 1;
    const float s0 = (float)S0, v0f = (float)v0;
#define OMC_LAUNCH_HES(V, SC)                                                                     \
    hipLaunchKernelGGL((heston_paths_kernel<V, SC>), dim3(grid_for((P + V - 1) / V)),             \
                       dim3(kBlock), 0, st, S, ld, P, n_steps, s0, v0f, c, k0, k1, stream,        \
                       pair_offset)
    if (scheme == 0) {
        if (vec == 4) OMC_LAUNCH_HES(4, 0);
        else if (vec == 2) OMC_LAUNCH_HES(2, 0);
        else OMC_LAUNCH_HES(1, 0);
    } else {
        if (vec == 4) OMC_LAUNCH_HES(4, 1);
        else if (vec == 2) OMC_LAUNCH_HES(2, 1);
        else OMC_LAUNCH_HES(1, 1);
    }
#undef OMC_LAUNCH_HES
    return hipGetLastError();
}

hipError_t launch_gbm_from_normals(hipStream_t st, float* S, int64_t ld, int64_t n_paths,
                                   int n_steps, double S0, double r, double sigma, double T,
                                   const float* Z, int64_t ldz, int antithetic)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    const float a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    const float b = (float)(sigma * sqrt(dt) * L2E);
    const int64_t P = antithetic ? n_paths / 2 : n_paths;
    if (P <= 0) return hipSuccess;
    if (antithetic)
        hipLaunchKernelGGL((gbm_from_normals_kernel<true>), dim3(grid_for(P)), dim3(kBlock), 0, st,
                           S, ld, P, n_steps, (float)S0, a, b, Z, ldz);
    else
        hipLaunchKernelGGL((gbm_from_normals_kernel<false>), dim3(grid_for(P)), dim3(kBlock), 0, st,
                           S, ld, P, n_steps, (float)S0, a, b, Z, ldz);
    return hipGetLastError();
}

hipError_t launch_heston_from_normals(hipStream_t st, float* S, int64_t ld, int64_t n_paths,
                                      int n_steps, double S0, double r, double T, double v0,
                                      double kappa, double theta, double xi, double rho,
                                      const float* Z1, const float* Z2, int64_t ldz, int scheme)
{
    const HestonC c = make_heston(r, T, n_steps, kappa, theta, xi, rho);
    const int64_t P = n_paths / 2;
    if (P <= 0) return hipSuccess;
    if (scheme == 0)
        hipLaunchKernelGGL((heston_from_normals_kernel<0>), dim3(grid_for(P)), dim3(kBlock), 0, st,
                           S, ld, P, n_steps, (float)S0, (float)v0, c, Z1, Z2, ldz);
    else
        hipLaunchKernelGGL((heston_from_normals_kernel<1>), dim3(grid_for(P)), dim3(kBlock), 0, st,
                           S, ld, P, n_steps, (float)S0, (float)v0, c, Z1, Z2, ldz);
    return hipGetLastError();
}

hipError_t launch_terminal(hipStream_t st, double* part, int* nblk_out, int model, int scheme,
                           int antithetic, int64_t n_paths, int n_steps, double S0, double K,
                           double r, double sigma, double T, double v0, double kappa, double theta,
                           double xi, double rho, int is_put, uint64_t seed, uint32_t stream,
                           uint64_t pair_offset)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    TermArgs a;
    a.P = (model == 0 && !antithetic) ? n_paths : n_paths / 2;
    a.n_steps = n_steps; a.is_put = is_put;
    a.s_init = (float)S0; a.v_init = (float)v0;
    a.a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    a.b = (float)(sigma * sqrt(dt) * L2E);
    a.hc = make_heston(r, T, n_steps, kappa, theta, xi, rho);
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.stream = stream;
    a.pair_offset = pair_offset; a.K = K; a.df = exp(-r * T); a.part = part;
    int64_t nb = (a.P + kBlock - 1) / kBlock;
    if (nb < 1) nb = 1;
    if (nb > kMaxLsmBlocks) nb = kMaxLsmBlocks;
    *nblk_out = (int)nb;
    const dim3 grid((unsigned)nb), block(kBlock);
    if (model == 0) {
        if (antithetic) hipLaunchKernelGGL((terminal_kernel<0, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((terminal_kernel<0, false>), grid, block, 0, st, a);
    } else if (scheme == 0) {
        hipLaunchKernelGGL((terminal_kernel<1, true>), grid, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((terminal_kernel<2, true>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_philox_kat(hipStream_t st, const uint32_t* in, uint32_t* out, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(philox_kat_kernel, dim3((n + 63) / 64), dim3(64), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_gbm_normals(hipStream_t st, float* Z, int64_t ldz, int64_t n_pairs, int n_steps,
                              uint64_t seed, uint32_t stream, uint64_t pair_offset)
{
    if (n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(gbm_normals_kernel, dim3(grid_for(n_pairs)), dim3(kBlock), 0, st, Z, ldz,
                       n_pairs, n_steps, (uint32_t)seed, (uint32_t)(seed >> 32), stream,
                       pair_offset);
    return hipGetLastError();
}

}  // namespace omc
