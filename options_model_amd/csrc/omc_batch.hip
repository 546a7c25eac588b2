// omc_batch.hip -- many small pricings in a handful of launches.
//
// The reference prices a value-vs-expiry curve as thousands of independent small problems
// (compute_curve_for_S0: options_model_3/options_model_3.py:697-713, Options_model.py:190-211,
// options_model_2.py:336-355; fanned out over processes at :1053-1056 / options_model_2_ui.py:87-133;
// GPU intent compute_multiple_S0_gpu_batch option_model_3_gpu.py:934-956 is a sequential loop).
// A 10k-path problem fills ~2 % of an MI355X, so here the batch dimension is blockIdx.z (y for
// the generators): every phase of EVERY problem goes in one launch sized for the largest
// problem, each workgroup reads its problem's argument block from a device table, and
// workgroups beyond a problem's own extent leave at once.  The kernel bodies are the ones of the
// single-problem path (omc_lsm_dev.h / omc_paths_dev.h): same arithmetic, same results.
#include "omc_batch.h"

#include <algorithm>

#include "omc_lsm_dev.h"
#include "omc_paths_dev.h"

namespace omc {

HestonC make_heston(double r, double T, int n_steps, double kappa, double theta, double xi, double rho);

// One per problem, built on the host, read-only on the device.
struct BatchProb {
    PathArgs path;
    TermArgs term;
    StepArgs step;
    Pass1Args p1;
    Pass2Args p2;
    FinalArgs fin;
    double* result;
    int fin_nblk;  // slots the finalize kernel adds up
};

// ------------------------------------------------------------------ batched entry points
template <int VEC, int GEN>  // GEN 0 GBM antithetic, 1 GBM plain, 2 Heston clamp, 3 full truncation, 4 calibrator
__global__ __launch_bounds__(kBlock) void paths_batch_kernel(const BatchProb* __restrict__ pr)
{
    const PathArgs g = pr[blockIdx.y].path;
    if constexpr (GEN == 0) gbm_paths_body<VEC, true>(g);
    else if constexpr (GEN == 1) gbm_paths_body<VEC, false>(g);
    else if constexpr (GEN == 2) heston_paths_body<VEC, 0>(g);
    else if constexpr (GEN == 3) heston_paths_body<VEC, 1>(g);
    else heston_paths_body<VEC, 2>(g);
}

template <int GEN>
__global__ __launch_bounds__(kBlock) void terminal_batch_kernel(const BatchProb* __restrict__ pr)
{
    const TermArgs a = pr[blockIdx.z].term;
    if constexpr (GEN == 0) terminal_body<0, true>(a);
    else if constexpr (GEN == 1) terminal_body<0, false>(a);
    else if constexpr (GEN == 2) terminal_body<1, true>(a);
    else if constexpr (GEN == 3) terminal_body<2, true>(a);
    else terminal_body<3, true>(a);
}

template <int SEM, int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lsm_step_batch_kernel(const BatchProb* __restrict__ pr, int t)
{
    StepArgs a = pr[blockIdx.z].step;
    a.t = t;
    lsm_step_body<SEM, VEC, BLOCK>(a, blockIdx.x, a.nblk);  // leaves at once when t > N or blockIdx.x >= nblk
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void lsm_pass1_batch_kernel(const BatchProb* __restrict__ pr)
{
    const Pass1Args a = pr[blockIdx.z].p1;
    lsm_pass1_body<VEC, 4>(a);
}

__global__ __launch_bounds__(kBlock) void lsm_reduce_pass1_batch_kernel(const BatchProb* __restrict__ pr)
{
    const Pass1Args& a = pr[blockIdx.z].p1;
    lsm_reduce_pass1_body(a.part1, pr[blockIdx.z].step.gmom, a.ntiles, a.N);
}

__global__ void lsm_solve_all_batch_kernel(const BatchProb* __restrict__ pr)
{
    const StepArgs& a = pr[blockIdx.z].step;
    lsm_solve_all_body(a.gmom, a.betas, a.N);
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void lsm_pass2_batch_kernel(const BatchProb* __restrict__ pr)
{
    const Pass2Args a = pr[blockIdx.z].p2;
    lsm_pass2_body<VEC, false>(a);
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void lsm_final_batch_kernel(const BatchProb* __restrict__ pr)
{
    const FinalArgs a = pr[blockIdx.z].fin;
    lsm_final_body<VEC>(a);
}

__global__ __launch_bounds__(kBlock) void lsm_finalize_batch_kernel(const BatchProb* __restrict__ pr,
                                                                    int with_moments)
{
    const BatchProb& p = pr[blockIdx.z];
    lsm_finalize_body(p.fin.part, p.step.gmom, p.result, p.fin_nblk, with_moments ? p.step.N : 0,
                      p.fin.pstride);
}

// ------------------------------------------------------------------ host side
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Layout {  // byte offsets of one problem's buffers inside the slab
    size_t S, sx, tex, ex, part, gmom, betas, part1, end;
    int64_t ld, ntiles;
    int nblk_sweep, nblk_blocks, pstride;
};

static bool batch_vec4(const BatchItem* items, int n)
{
    for (int i = 0; i < n; ++i) {
        const int64_t M = items[i].n_paths;
        const int64_t P = (items[i].model == 0 && !items[i].antithetic) ? M : M / 2;
        if ((P % 4) != 0 || (M % 4) != 0) return false;
    }
    return true;
}

static Layout plan(const BatchItem& it, size_t base, bool american, bool two_pass, bool vec4)
{
    Layout L{};
    const int64_t M = it.n_paths;
    const int N = it.n_steps;
    L.ld = (M + 63) / 64 * 64;
    L.nblk_sweep = lsm_sweep_blocks(M);
    L.nblk_blocks = lsm_step_blocks(M);
    L.pstride = std::max(L.nblk_sweep, L.nblk_blocks);
    if (!american) {  // terminal-only kernel: same grid as the single-problem launcher
        const int64_t P = (it.model == 0 && !it.antithetic) ? M : M / 2;
        L.pstride = (int)std::max<int64_t>(1, std::min<int64_t>((P + kBlock - 1) / kBlock, kMaxLsmBlocks));
    }
    const int64_t per_wave = 64 * (vec4 ? 4 : 1) * 4;  // pass 1: lanes x VEC x 4 tiles per wave
    L.ntiles = (M + per_wave - 1) / per_wave;
    size_t o = align_up(base, 256);
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    if (american) {
        L.S = take(sizeof(float) * (size_t)L.ld * (size_t)(N + 1));
        L.sx = take(sizeof(float) * (size_t)M);
        L.tex = take(sizeof(int32_t) * (size_t)M);
        L.ex = take((size_t)M);
        L.gmom = take(sizeof(double) * 8 * (size_t)(N + 1));
        L.betas = take(sizeof(double) * 4 * (size_t)(N + 1));
        if (two_pass) L.part1 = take(sizeof(double) * 8 * (size_t)(N + 1) * (size_t)L.ntiles);
    }
    L.part = take(sizeof(double) * 2 * 8 * (size_t)L.pstride);
    L.end = o;
    return L;
}

size_t batch_slab_bytes(const BatchItem* items, int n, bool american, bool two_pass)
{
    size_t o = 0;
    const bool v4 = batch_vec4(items, n);
    for (int i = 0; i < n; ++i) o = plan(items[i], o, american, two_pass, v4).end;
    return o;
}

size_t batch_table_bytes(int n) { return sizeof(BatchProb) * (size_t)n; }

size_t batch_discount_doubles(const BatchItem* items, int n)
{
    size_t k = 0;
    for (int i = 0; i < n; ++i) k += (size_t)items[i].n_steps + 1;
    return k;
}

// Fills the host copy of the problem table (`table_host`, batch_table_bytes) and the host
// discount staging area; returns launch extents through `ext`.
void batch_build(const BatchItem* items, int n, bool american, bool two_pass, char* slab,
                 double* results_dev, double* disc_dev, void* table_host, double* disc_host,
                 BatchExtents* ext)
{
    BatchProb* tab = (BatchProb*)table_host;
    const double L2E = 1.4426950408889634074;
    size_t o = 0, dk = 0;
    BatchExtents e{};
    e.vec4 = batch_vec4(items, n) ? 1 : 0;
    for (int i = 0; i < n; ++i) {
        const BatchItem& it = items[i];
        const Layout L = plan(it, o, american, two_pass, e.vec4 != 0);
        o = L.end;
        BatchProb& p = tab[i];
        p = BatchProb{};
        const int64_t M = it.n_paths;
        const int N = it.n_steps;
        const int64_t P = (it.model == 0 && !it.antithetic) ? M : M / 2;
        const double dt = it.T / N;
        // generator
        p.path.S = american ? (float*)(slab + L.S) : nullptr;
        p.path.ld = L.ld; p.path.P = P; p.path.n_steps = N;
        p.path.s_init = (float)it.S0; p.path.v_init = (float)it.v0;
        p.path.a = (float)((it.r - 0.5 * it.sigma * it.sigma) * dt * L2E);
        p.path.b = (float)(it.sigma * sqrt(dt) * L2E);
        p.path.hc = make_heston(it.r, it.T, N, it.kappa, it.theta, it.xi, it.rho);
        p.path.k0 = (uint32_t)it.seed; p.path.k1 = (uint32_t)(it.seed >> 32);
        p.path.stream = (uint32_t)it.stream; p.path.pair_offset = it.pair_offset;
        double* part = (double*)(slab + L.part);
        p.result = results_dev + 8 * (size_t)i;
        if (!american) {
            TermArgs& t = p.term;
            t.P = P; t.n_steps = N; t.is_put = it.is_put; t.s_init = p.path.s_init; t.a = p.path.a;
            t.b = p.path.b; t.v_init = p.path.v_init; t.hc = p.path.hc; t.k0 = p.path.k0; t.k1 = p.path.k1;
            t.stream = p.path.stream; t.pair_offset = it.pair_offset; t.K = it.K; t.df = exp(-it.r * it.T);
            t.part = part;
            int64_t nb = (P + kBlock - 1) / kBlock;
            nb = std::max<int64_t>(1, std::min<int64_t>(nb, L.pstride));
            t.nblk = (int)nb; t.pstride = L.pstride;
            p.fin.part = part; p.fin.pstride = L.pstride; p.fin_nblk = (int)nb;
            e.term_blocks = std::max(e.term_blocks, (int)nb);
            continue;
        }
        // all discount tables are contiguous (one H2D copy): problem i owns [dk, dk + N]
        double* D = disc_dev + dk;
        for (int k = 0; k <= N; ++k) disc_host[dk + (size_t)k] = exp(-it.r * dt * (double)k);
        dk += (size_t)N + 1;
        StepArgs& s = p.step;
        s.S = p.path.S; s.ld = L.ld; s.M = M; s.N = N; s.is_put = it.is_put; s.K = it.K; s.invK = 1.0 / it.K;
        s.sx = (float*)(slab + L.sx); s.tex = (int32_t*)(slab + L.tex); s.ex = (uint8_t*)(slab + L.ex); s.D = D;
        s.part = part;
        s.gmom = (double*)(slab + L.gmom); s.betas = (double*)(slab + L.betas);
        s.t = 0; s.nblk = L.nblk_sweep; s.external = 0; s.pstride = L.pstride; s.gstride = 8; s.cont = nullptr; s.ldc = 0; s.dbg = nullptr;
        Pass1Args& a1 = p.p1;
        a1.S = s.S; a1.ld = L.ld; a1.M = M; a1.N = N; a1.is_put = it.is_put; a1.K = it.K; a1.invK = s.invK;
        a1.D = D; a1.part1 = two_pass ? (double*)(slab + L.part1) : nullptr; a1.ntiles = L.ntiles; a1.tchunk = 16;
        Pass2Args& a2 = p.p2;
        a2.S = s.S; a2.ld = L.ld; a2.M = M; a2.N = N; a2.is_put = it.is_put; a2.K = it.K; a2.invK = s.invK;
        a2.D = D; a2.betas = s.betas; a2.sx = s.sx; a2.tex = s.tex; a2.part = part;
        a2.nblk = L.nblk_blocks; a2.pstride = L.pstride;
        FinalArgs& f = p.fin;
        f.sx = s.sx; f.tex = s.tex; f.M = M; f.N = N; f.is_put = it.is_put; f.tval = it.semantics == 1 ? 0 : 1;
        f.ex = it.semantics == 0 ? s.ex : nullptr; f.SN = s.S + (int64_t)N * L.ld; f.fill_state = 0;
        f.K = it.K; f.D = D; f.part = part; f.nblk = L.nblk_blocks; f.pstride = L.pstride;
        p.fin_nblk = L.nblk_blocks;
        e.max_steps = std::max(e.max_steps, N);
        e.path_blocks = std::max<int64_t>(e.path_blocks, (P + kBlock - 1) / kBlock);
        e.sweep_blocks = std::max(e.sweep_blocks, L.nblk_sweep);
        e.block_blocks = std::max(e.block_blocks, L.nblk_blocks);
        e.tile_blocks = std::max<int64_t>(e.tile_blocks, (L.ntiles + 3) / 4);
    }
    *ext = e;
}

hipError_t batch_paths(hipStream_t st, const void* table_dev, int n, const BatchExtents& e, int gen)
{
    const BatchProb* pr = (const BatchProb*)table_dev;
    const int vec = e.vec4 ? 4 : 1;
    const dim3 grid((unsigned)((e.path_blocks + vec - 1) / vec), (unsigned)n), block(kBlock);
#define OMC_BP(V, G) hipLaunchKernelGGL((paths_batch_kernel<V, G>), grid, block, 0, st, pr)
    if (vec == 4) {
        if (gen == 0) OMC_BP(4, 0); else if (gen == 1) OMC_BP(4, 1); else if (gen == 2) OMC_BP(4, 2);
        else if (gen == 3) OMC_BP(4, 3); else OMC_BP(4, 4);
    } else {
        if (gen == 0) OMC_BP(1, 0); else if (gen == 1) OMC_BP(1, 1); else if (gen == 2) OMC_BP(1, 2);
        else if (gen == 3) OMC_BP(1, 3); else OMC_BP(1, 4);
    }
#undef OMC_BP
    return hipGetLastError();
}

hipError_t batch_lsm(hipStream_t st, const void* table_dev, int n, const BatchExtents& e, int semantics)
{
    const BatchProb* pr = (const BatchProb*)table_dev;
    const unsigned z = (unsigned)n;
    const int Nmax = e.max_steps;
    if (semantics == 2) {
        if (Nmax >= 2) {
            const dim3 g1((unsigned)e.tile_blocks, (unsigned)((Nmax - 1 + 15) / 16), z);
            if (e.vec4) hipLaunchKernelGGL((lsm_pass1_batch_kernel<4>), g1, dim3(kBlock), 0, st, pr);
            else hipLaunchKernelGGL((lsm_pass1_batch_kernel<1>), g1, dim3(kBlock), 0, st, pr);
            hipLaunchKernelGGL(lsm_reduce_pass1_batch_kernel, dim3(Nmax - 1, 8, z), dim3(kBlock), 0, st, pr);
            hipLaunchKernelGGL(lsm_solve_all_batch_kernel, dim3((Nmax + 255) / 256, 1, z), dim3(256), 0, st, pr);
        }
        const size_t dyn = sizeof(double) * 4 * (size_t)(Nmax + 1);
        const dim3 g2((unsigned)e.block_blocks, 1, z);
        if (e.vec4) hipLaunchKernelGGL((lsm_pass2_batch_kernel<4>), g2, dim3(kBlock), dyn, st, pr);
        else hipLaunchKernelGGL((lsm_pass2_batch_kernel<1>), g2, dim3(kBlock), dyn, st, pr);
    } else {
        const bool big = lsm_step_block_threads() == 1024;
        const dim3 gs((unsigned)e.sweep_blocks, 1, z), bs(big ? 1024 : 512);
        const size_t dyn = semantics == 1 ? sizeof(double) * (size_t)(Nmax + 1) : 0;
#define OMC_BSTEP(SEM, VEC)                                                                              \
    do {                                                                                                 \
        if (big) hipLaunchKernelGGL((lsm_step_batch_kernel<SEM, VEC, 1024>), gs, bs, dyn, st, pr, t);    \
        else hipLaunchKernelGGL((lsm_step_batch_kernel<SEM, VEC, 512>), gs, bs, dyn, st, pr, t);         \
    } while (0)
        for (int t = Nmax; t >= 1; --t) {
            if (semantics == 0) {
                if (e.vec4) OMC_BSTEP(0, 4); else OMC_BSTEP(0, 1);
            } else {
                if (e.vec4) OMC_BSTEP(1, 4); else OMC_BSTEP(1, 1);
            }
        }
#undef OMC_BSTEP
        const dim3 gf((unsigned)e.block_blocks, 1, z);
        if (e.vec4) hipLaunchKernelGGL((lsm_final_batch_kernel<4>), gf, dim3(kBlock), 0, st, pr);
        else hipLaunchKernelGGL((lsm_final_batch_kernel<1>), gf, dim3(kBlock), 0, st, pr);
    }
    hipLaunchKernelGGL(lsm_finalize_batch_kernel, dim3(1, 1, z), dim3(kBlock), 0, st, pr, 1);
    return hipGetLastError();
}

hipError_t batch_terminal(hipStream_t st, const void* table_dev, int n, const BatchExtents& e, int gen)
{
    const BatchProb* pr = (const BatchProb*)table_dev;
    const dim3 grid((unsigned)e.term_blocks, 1, (unsigned)n);
    if (gen == 0) hipLaunchKernelGGL((terminal_batch_kernel<0>), grid, dim3(kBlock), 0, st, pr);
    else if (gen == 1) hipLaunchKernelGGL((terminal_batch_kernel<1>), grid, dim3(kBlock), 0, st, pr);
    else if (gen == 2) hipLaunchKernelGGL((terminal_batch_kernel<2>), grid, dim3(kBlock), 0, st, pr);
    else if (gen == 3) hipLaunchKernelGGL((terminal_batch_kernel<3>), grid, dim3(kBlock), 0, st, pr);
    else hipLaunchKernelGGL((terminal_batch_kernel<4>), grid, dim3(kBlock), 0, st, pr);
    hipLaunchKernelGGL(lsm_finalize_batch_kernel, dim3(1, 1, (unsigned)n), dim3(kBlock), 0, st, pr, 0);
    return hipGetLastError();
}

}  // namespace omc
