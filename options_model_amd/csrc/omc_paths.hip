// omc_paths.hip -- path generation kernels for gfx950 (MI355X).
//
// Replaces the reference's per-step numpy/torch op chains
//   GBM    options_model_3/options_model_3.py:473-480   (GPU intent: option_model_3_gpu.py:117-185)
//   Heston options_model_3/options_model_3.py:211-251   (GPU intent: option_model_3_gpu.py:187-248)
// with ONE launch each: a thread owns VEC antithetic pairs, keeps S (and the Heston
// variance) in registers for the whole time loop, draws its normals from a counter RNG
// (no state, no reads) and streams the [step][path] matrix out with 4*VEC-byte stores.
// The kernel is pure HBM write traffic: (n_steps+1) * n_paths * 4 bytes.
#include "omc_paths_dev.h"

namespace omc {

// ------------------------------------------------------------------ __global__ entry points
template <int VEC, bool ANTI>
__global__ __launch_bounds__(kBlock) void gbm_paths_kernel(PathArgs g) { gbm_paths_body<VEC, ANTI>(g); }

template <int VEC, int SCHEME>
__global__ __launch_bounds__(kBlock) void heston_paths_kernel(PathArgs g) { heston_paths_body<VEC, SCHEME>(g); }

template <int MODEL, bool ANTI>
__global__ __launch_bounds__(kBlock) void terminal_kernel(TermArgs a) { terminal_body<MODEL, ANTI>(a); }

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
        heston_pair_step<SCHEME>(c, z1, z2, s, va, sa, vb);
        S[(int64_t)t * ld + p] = s;
        S[(int64_t)t * ld + p + P] = sa;
    }
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
HestonC make_heston(double r, double T, int n_steps, double kappa, double theta,
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
    c.sqdt = (float)sqrt(dt);
    c.rdt = (float)(r * dt);
    c.xi_sqdt = (float)(xi * sqrt(dt));
    c.l2e_sqdt = (float)(L2E * sqrt(dt));
    return c;
}

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
    PathArgs g{};
    g.S = S; g.ld = ld; g.P = P; g.n_steps = n_steps; g.s_init = (float)S0; g.a = a; g.b = b;
    g.k0 = k0; g.k1 = k1; g.stream = stream; g.pair_offset = pair_offset;
#define OMC_LAUNCH_GBM(V, A)                                                                       \
    hipLaunchKernelGGL((gbm_paths_kernel<V, A>), dim3(grid_for((P + V - 1) / V)), dim3(kBlock), 0, \
                       st, g)
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
    PathArgs g{};
    g.S = S; g.ld = ld; g.P = P; g.n_steps = n_steps; g.s_init = (float)S0; g.v_init = (float)v0;
    g.hc = c; g.k0 = k0; g.k1 = k1; g.stream = stream; g.pair_offset = pair_offset;
#define OMC_LAUNCH_HES(V, SC)                                                                     \
    hipLaunchKernelGGL((heston_paths_kernel<V, SC>), dim3(grid_for((P + V - 1) / V)),             \
                       dim3(kBlock), 0, st, g)
    if (scheme == 0) {
        if (vec == 4) OMC_LAUNCH_HES(4, 0);
        else if (vec == 2) OMC_LAUNCH_HES(2, 0);
        else OMC_LAUNCH_HES(1, 0);
    } else if (scheme == 1) {
        if (vec == 4) OMC_LAUNCH_HES(4, 1);
        else if (vec == 2) OMC_LAUNCH_HES(2, 1);
        else OMC_LAUNCH_HES(1, 1);
    } else {
        if (vec == 4) OMC_LAUNCH_HES(4, 2);
        else if (vec == 2) OMC_LAUNCH_HES(2, 2);
        else OMC_LAUNCH_HES(1, 2);
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
    else if (scheme == 1)
        hipLaunchKernelGGL((heston_from_normals_kernel<1>), dim3(grid_for(P)), dim3(kBlock), 0, st,
                           S, ld, P, n_steps, (float)S0, (float)v0, c, Z1, Z2, ldz);
    else
        hipLaunchKernelGGL((heston_from_normals_kernel<2>), dim3(grid_for(P)), dim3(kBlock), 0, st,
                           S, ld, P, n_steps, (float)S0, (float)v0, c, Z1, Z2, ldz);
    return hipGetLastError();
}

// ------------------------------------------------------------------ calibrator inner loop
// HestonPricer.price_options_batch (heston_calibration.py:283-312): one simulation per expiry,
// then one mean payoff per strike.  Terminal spots of all paths go to a small buffer (4 bytes
// per path, no path matrix); a second launch reduces one strike per workgroup over it.
// one antithetic pair from t = 0 to the expiry: the terminal spots of both partners (shared by the one-expiry kernel and
// the surface kernel, so that a surface's expiry has the bits of its own omc_heston_price_strikes call)
template <int SCHEME>
__device__ __forceinline__ void heston_terminal_pair(const HestonC& hc, int n_steps, float s_init, float v_init, uint64_t pair,
                                                     uint32_t stream, uint32_t k0, uint32_t k1, float& s_out, float& sa_out)
{
    float s = s_init, sa = s_init, va = v_init, vb = v_init, z[4];
    for (int t = 0; t < n_steps; ++t) {
        const int i = t & 1;
        if (i == 0) normals4(pair, (uint32_t)(t >> 1), stream, k0, k1, z);
        heston_pair_step<SCHEME>(hc, z[2 * i], z[2 * i + 1], s, va, sa, vb);
    }
    s_out = s;
    sa_out = sa;
}

template <int SCHEME>
__global__ __launch_bounds__(kBlock) void heston_terminal_store_kernel(float* __restrict__ ST, PathArgs g)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= g.P) return;
    float s, sa;
    heston_terminal_pair<SCHEME>(g.hc, g.n_steps, g.s_init, g.v_init, g.pair_offset + (uint64_t)p, g.stream, g.k0, g.k1, s, sa);
    ST[p] = s;
    ST[p + g.P] = sa;
}

// A whole quote surface in ONE launch set (HestonPricer.price_options_batch, heston_calibration.py:283-312: the calibrator's
// objective simulates every distinct expiry and averages every strike of it): the expiry on grid.y -- its own dt-dependent
// constants and Philox sub-stream from a small device table --, terminal spots into ST[expiry][ldst]; then one workgroup
// per QUOTE reduces its strike over its expiry's row, in payoff_means_kernel's order.
struct SurfaceExpiry {
    HestonC hc;
    uint32_t stream, pad[2];
};
static_assert(sizeof(SurfaceExpiry) == 64, "omc::heston_surface_table_bytes");

template <int SCHEME>
__global__ __launch_bounds__(kBlock) void heston_terminal_surface_kernel(float* __restrict__ ST, int64_t ldst, PathArgs g,
                                                                         const SurfaceExpiry* __restrict__ tab)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= g.P) return;
    const SurfaceExpiry e = tab[blockIdx.y];
    float s, sa;
    heston_terminal_pair<SCHEME>(e.hc, g.n_steps, g.s_init, g.v_init, g.pair_offset + (uint64_t)p, e.stream, g.k0, g.k1, s, sa);
    float* row = ST + (int64_t)blockIdx.y * ldst;
    row[p] = s;
    row[p + g.P] = sa;
}

// {sum, sumsq} of one strike's payoffs over one row of terminal spots, in two levels (round 6: one workgroup per strike
// walking all 100,000 spots was 105 us of dependent loads and float64 adds on ten CUs -- four fifths of a calibrator
// evaluation): workgroup (chunk, quote) sums kPayChunk spots -- thread j adds spots j, j + 256, ... of the chunk in order,
// then the workgroup tree --, then one thread per quote adds the chunks' sums in chunk order.  The summation order of a
// quote is fixed by (M, kPayChunk, block size) alone: the single-expiry call and the surface call share it bit for bit.
constexpr int kPayChunk = 4096;

__device__ __forceinline__ void payoff_chunk_body(const float* __restrict__ ST, int64_t M, double k, int is_put,
                                                  const int chunk, double* __restrict__ out2)
{
    __shared__ double red[kNQ * kRedStride];
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const int64_t lo = (int64_t)chunk * kPayChunk, hi = lo + kPayChunk < M ? lo + kPayChunk : M;
    float sv[kPayChunk / kBlock];
#pragma unroll
    for (int i = 0; i < kPayChunk / kBlock; ++i) {  // every load of the chunk in flight before the first use
        const int64_t j = lo + threadIdx.x + (int64_t)i * kBlock;
        sv[i] = j < hi ? ST[j] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < kPayChunk / kBlock; ++i) {
        const int64_t j = lo + threadIdx.x + (int64_t)i * kBlock;
        double p = payoff_d(sv[i], k, is_put);
        p = (j < hi && p > 0.0) ? p : 0.0;
        acc[0] += p;
        acc[1] += p * p;
    }
    const double s = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0 && (threadIdx.x >> 3) < 2) out2[threadIdx.x >> 3] = s;
}

// part[quote][chunk][2]; grid (quotes, chunks): the quote on x (any number of them), the chunk on y (<= 65,535: 2.7e8 paths)
__global__ __launch_bounds__(kBlock) void payoff_partial_kernel(const float* __restrict__ ST, int64_t M,
                                                                const double* __restrict__ K, int is_put,
                                                                double* __restrict__ part)
{
    payoff_chunk_body(ST, M, K[blockIdx.x], is_put, (int)blockIdx.y, part + 2 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y));
}

__global__ __launch_bounds__(kBlock) void payoff_partial_surface_kernel(const float* __restrict__ ST, int64_t ldst, int64_t M,
                                                                        const double* __restrict__ K,
                                                                        const int32_t* __restrict__ expiry_of, int is_put,
                                                                        double* __restrict__ part)
{
    payoff_chunk_body(ST + (int64_t)expiry_of[blockIdx.x] * ldst, M, K[blockIdx.x], is_put, (int)blockIdx.y,
                      part + 2 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y));
}

// out[quote][2] = the chunks' sums added in chunk order
__global__ __launch_bounds__(kBlock) void payoff_final_kernel(const double* __restrict__ part, int nchunks, int n_quotes,
                                                              double* __restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;  // (quote, component)
    if (i >= 2 * n_quotes) return;
    const double* p = part + (size_t)(i >> 1) * nchunks * 2 + (i & 1);
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += p[2 * c];
    out[i] = s;
}

hipError_t launch_heston_terminal_store(hipStream_t st, float* ST, int64_t n_paths, int n_steps,
                                        double S0, double r, double T, double v0, double kappa,
                                        double theta, double xi, double rho, uint64_t seed,
                                        uint32_t stream, uint64_t pair_offset, int scheme)
{
    PathArgs g{};
    g.P = n_paths / 2; g.n_steps = n_steps; g.s_init = (float)S0; g.v_init = (float)v0;
    g.hc = make_heston(r, T, n_steps, kappa, theta, xi, rho);
    g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32); g.stream = stream; g.pair_offset = pair_offset;
    if (g.P <= 0) return hipSuccess;
    const dim3 grid(grid_for(g.P)), block(kBlock);
    if (scheme == 0) hipLaunchKernelGGL((heston_terminal_store_kernel<0>), grid, block, 0, st, ST, g);
    else if (scheme == 1) hipLaunchKernelGGL((heston_terminal_store_kernel<1>), grid, block, 0, st, ST, g);
    else hipLaunchKernelGGL((heston_terminal_store_kernel<2>), grid, block, 0, st, ST, g);
    return hipGetLastError();
}

size_t heston_surface_table_bytes(int n_expiries) { return sizeof(SurfaceExpiry) * (size_t)(n_expiries > 0 ? n_expiries : 0); }

// the per-expiry table image (host): constants of each expiry's time step and its Philox sub-stream
void heston_surface_fill_table(void* tab_host, int n_steps, double r, const double* T_host, const uint32_t* stream_host,
                               int n_expiries, double kappa, double theta, double xi, double rho)
{
    SurfaceExpiry* h = (SurfaceExpiry*)tab_host;
    for (int e = 0; e < n_expiries; ++e) {
        h[e] = SurfaceExpiry{};
        h[e].hc = make_heston(r, T_host[e], n_steps, kappa, theta, xi, rho);
        h[e].stream = stream_host[e];
    }
}

// (tab_dev: the table image of heston_surface_fill_table, already on the device or queued ahead on `st`)
hipError_t launch_heston_terminal_surface(hipStream_t st, float* ST, int64_t ldst, int64_t n_paths, int n_steps, double S0,
                                          int n_expiries, double v0, uint64_t seed, uint64_t pair_offset, int scheme,
                                          const void* tab_dev)
{
    PathArgs g{};
    g.P = n_paths / 2; g.n_steps = n_steps; g.s_init = (float)S0; g.v_init = (float)v0;
    g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32); g.pair_offset = pair_offset;
    if (g.P <= 0 || n_expiries <= 0) return hipSuccess;
    const dim3 grid(grid_for(g.P), (unsigned)n_expiries), block(kBlock);
    const SurfaceExpiry* tab = (const SurfaceExpiry*)tab_dev;
    if (scheme == 0) hipLaunchKernelGGL((heston_terminal_surface_kernel<0>), grid, block, 0, st, ST, ldst, g, tab);
    else if (scheme == 1) hipLaunchKernelGGL((heston_terminal_surface_kernel<1>), grid, block, 0, st, ST, ldst, g, tab);
    else hipLaunchKernelGGL((heston_terminal_surface_kernel<2>), grid, block, 0, st, ST, ldst, g, tab);
    return hipGetLastError();
}

size_t payoff_partial_bytes(int64_t n_paths, int n_quotes)
{
    const int64_t nchunks = (n_paths + kPayChunk - 1) / kPayChunk;
    return sizeof(double) * 2 * (size_t)(nchunks > 0 ? nchunks : 1) * (size_t)(n_quotes > 0 ? n_quotes : 0);
}

hipError_t launch_payoff_means_surface(hipStream_t st, const float* ST, int64_t ldst, int64_t n_paths, const double* K_dev,
                                       const int32_t* expiry_of_dev, int n_quotes, int is_put, double* part_dev, double* out_dev)
{
    if (n_quotes <= 0) return hipSuccess;
    const int nchunks = (int)((n_paths + kPayChunk - 1) / kPayChunk);
    hipLaunchKernelGGL(payoff_partial_surface_kernel, dim3(n_quotes, nchunks), dim3(kBlock), 0, st, ST, ldst, n_paths, K_dev,
                       expiry_of_dev, is_put, part_dev);
    hipLaunchKernelGGL(payoff_final_kernel, dim3((2 * n_quotes + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       (const double*)part_dev, nchunks, n_quotes, out_dev);
    return hipGetLastError();
}

hipError_t launch_payoff_means(hipStream_t st, const float* ST, int64_t n_paths, const double* K_dev,
                               int n_strikes, int is_put, double* part_dev, double* out_dev)
{
    if (n_strikes <= 0) return hipSuccess;
    const int nchunks = (int)((n_paths + kPayChunk - 1) / kPayChunk);
    hipLaunchKernelGGL(payoff_partial_kernel, dim3(n_strikes, nchunks), dim3(kBlock), 0, st, ST, n_paths, K_dev, is_put, part_dev);
    hipLaunchKernelGGL(payoff_final_kernel, dim3((2 * n_strikes + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       (const double*)part_dev, nchunks, n_strikes, out_dev);
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
    a.nblk = (int)nb; a.pstride = kMaxLsmBlocks;
    const dim3 grid((unsigned)nb), block(kBlock);
    if (model == 0) {
        if (antithetic) hipLaunchKernelGGL((terminal_kernel<0, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((terminal_kernel<0, false>), grid, block, 0, st, a);
    } else if (scheme == 0) {
        hipLaunchKernelGGL((terminal_kernel<1, true>), grid, block, 0, st, a);
    } else if (scheme == 1) {
        hipLaunchKernelGGL((terminal_kernel<2, true>), grid, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((terminal_kernel<3, true>), grid, block, 0, st, a);
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
