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

#include "omc_contnet_dev.h"
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

// ------------------------------------------------------------------ batched per-step ContNet flow
// The regressor the reference's v1 / v2 pricers really use (a fresh ContNet per time step, omc_contnet.hip) for
// MANY pricings at once: what their curve entry points run (Options_model.py:190-211, options_model_2.py:336-355:
// 1,620 pricings of 10k paths for the UI's default job).  Problem index on the grid for every kernel of the chain;
// the size of a step's regression set never leaves the device (the trainer's launches are sized for the largest
// problem and its workgroups read the set size from the step's header; surplus workgroups exit), so a whole batch
// runs without a single host read-back between its first and its last launch.  Bodies are the single-pricing ones:
// every problem ends with the bits of its own omc_price_american_contnet call.
struct CnProb {
    int32_t* cnt;
    double* s1;
    double* s2;
    int64_t* offs;
    double* hdr;   // n, mean, 1/std, std of the current step's set; [4] = running sum of the set sizes
    float* data;   // [M][8] trainer rows (capacity: every path)
    float* cont;   // [M]
    float* params; // padded net: params | m | v
    float* m;
    float* v;
    float* wt;     // the hidden-to-hidden connection transposed (the trainer's forward product reads it)
    int nblk, h, H, np;
    uint32_t k0, k1;
};

// false: problem `z` has no work at loop step t (its sweep is shorter, or t is its maturity)
__device__ __forceinline__ bool cn_args_at(const BatchProb& pr, const CnProb& c, int t, CnArgs* a)
{
    const StepArgs& s = pr.step;
    if (t < 1 || t >= s.N) return false;
    a->St = s.S + (int64_t)t * s.ld;
    a->SN = s.S + (int64_t)s.N * s.ld;
    a->live = s.live;
    a->M = s.M;
    a->K = s.K;
    a->Dt = s.D[s.N - t];
    a->is_put = s.is_put;
    a->nblk = c.nblk;
    a->cnt = c.cnt; a->s1 = c.s1; a->s2 = c.s2; a->offs = c.offs; a->hdr = c.hdr; a->data = c.data; a->cont = c.cont;
    return true;
}

__global__ __launch_bounds__(kCnBlock) void cn_count_batch_kernel(const BatchProb* __restrict__ pr,
                                                                 const CnProb* __restrict__ cp, int t)
{
    CnArgs a;
    if (!cn_args_at(pr[blockIdx.y], cp[blockIdx.y], t, &a) || (int)blockIdx.x >= a.nblk) return;
    cn_count_body(a);
}

// one workgroup per problem: scan of the counts -> row offsets and the set's header; then the step's FRESH net
// (cn_init_one; Adam moments zeroed) together with the transposed copy of its hidden-to-hidden connection
__global__ __launch_bounds__(1024) void cn_scan_init_batch_kernel(const BatchProb* __restrict__ pr,
                                                                  const CnProb* __restrict__ cp, int t)
{
    const CnProb& c = cp[blockIdx.y];
    CnArgs a;
    if (!cn_args_at(pr[blockIdx.y], c, t, &a)) return;
    cn_scan_body(a);
    if (threadIdx.x == 0) c.hdr[4] += c.hdr[0];
    CnInitArgs ia;
    ia.params = c.params; ia.m = c.m; ia.v = c.v;
    ia.H = c.H; ia.h = c.h; ia.np = c.np;
    ia.k0 = c.k0; ia.k1 = c.k1; ia.t = (uint32_t)t;
    const int H = c.H;
    for (int i = threadIdx.x; i < c.np; i += 1024) {
        cn_init_one(ia, i);
        const int e = i - H * 8;
        if (e >= 0 && e < H * H) c.wt[(size_t)(e % H) * H + e / H] = c.params[i];
    }
}

__global__ __launch_bounds__(kCnBlock) void cn_rows_batch_kernel(const BatchProb* __restrict__ pr,
                                                                const CnProb* __restrict__ cp, int t)
{
    CnArgs a;
    if (!cn_args_at(pr[blockIdx.y], cp[blockIdx.y], t, &a) || (int)blockIdx.x >= a.nblk) return;
    cn_rows_body(a);
}

template <int H>
__global__ __launch_bounds__(kCnBlock) void cn_forward_batch_kernel(const BatchProb* __restrict__ pr,
                                                                   const CnProb* __restrict__ cp, int t)
{
    const CnProb& c = cp[blockIdx.y];
    CnFwdArgs fa;
    if (!cn_args_at(pr[blockIdx.y], c, t, &fa.c) || (int64_t)blockIdx.x * kCnBlock >= fa.c.M) return;
    fa.params = c.params;
    fa.h = c.h;
    cn_forward_body<H>(fa);
}

// result slot 4 = sum over the steps of the regression-set sizes (the rows the nets were trained on)
__global__ void cn_total_batch_kernel(const BatchProb* __restrict__ pr, const CnProb* __restrict__ cp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pr[i].result[4] = cp[i].hdr[4];
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
        L.ex = take(sizeof(float) * (size_t)M);  // the per-step reference flow's `live` row
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
        s.sx = (float*)(slab + L.sx); s.tex = (int32_t*)(slab + L.tex); s.live = (float*)(slab + L.ex); s.D = D;
        s.part = part;
        s.gmom = (double*)(slab + L.gmom); s.betas = (double*)(slab + L.betas);
        s.t = 0; s.nblk = L.nblk_sweep; s.external = 0; s.pstride = L.pstride; s.gstride = 8; s.cont = nullptr; s.ldc = 0;
        Pass1Args& a1 = p.p1;
        a1.S = s.S; a1.ld = L.ld; a1.M = M; a1.N = N; a1.is_put = it.is_put; a1.K = it.K; a1.invK = s.invK;
        a1.D = D; a1.part1 = two_pass ? (double*)(slab + L.part1) : nullptr; a1.ntiles = L.ntiles; a1.tchunk = 16;
        Pass2Args& a2 = p.p2;
        a2.S = s.S; a2.ld = L.ld; a2.M = M; a2.N = N; a2.is_put = it.is_put; a2.K = it.K; a2.invK = s.invK;
        a2.D = D; a2.betas = s.betas; a2.sx = s.sx; a2.tex = s.tex; a2.part = part;
        a2.nblk = L.nblk_blocks; a2.pstride = L.pstride;
        FinalArgs& f = p.fin;
        f.sx = s.sx; f.tex = s.tex; f.M = M; f.N = N; f.is_put = it.is_put; f.tval = it.semantics == 1 ? 0 : 1;
        f.live = it.semantics == 0 ? s.live : nullptr; f.fill_state = 0;
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
            hipLaunchKernelGGL(lsm_reduce_pass1_batch_kernel, dim3(Nmax - 1, 1, z), dim3(kBlock), 0, st, pr);
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

// ---- batched per-step ContNet flow: host side
size_t batch_cn_table_bytes(int n) { return sizeof(CnProb) * (size_t)n; }

static size_t cn_plan(const BatchItem& it, int H, size_t base, size_t* off /*[11]*/)
{
    const int64_t M = it.n_paths;
    const size_t nb = (size_t)((M + kCnSpan - 1) / kCnSpan);
    const int np = H * 8 + H * H + H + H + 1;
    const size_t tiles = (size_t)((M + 31) / 32);
    const size_t pstride = (size_t)((np + 1 + 63) / 64 * 64);  // the tile trainer's partial stride (omc_mlp.hip)
    size_t o = align_up(base, 256);
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    off[0] = take(4 * (nb + 2));            // cnt
    off[1] = take(8 * nb);                  // s1
    off[2] = take(8 * nb);                  // s2
    off[3] = take(8 * (nb + 1));            // offs
    off[4] = take(sizeof(float) * 8 * (size_t)M);  // data
    off[5] = take(sizeof(float) * (size_t)M);      // cont
    off[6] = take(sizeof(float) * 3 * (size_t)np); // params | m | v
    off[7] = take(sizeof(float) * (size_t)H * H);  // wt
    off[8] = take(sizeof(float) * tiles * pstride); // gradient partials, one per tile
    return o;
}

// bytes of the second slab: [n][8] doubles of headers first (cleared before every batch), then per-problem buffers
size_t batch_cn_slab_bytes(const BatchItem* items, int n, int hidden)
{
    const int H = cn_padded_width(hidden);
    size_t o = align_up(sizeof(double) * 8 * (size_t)n, 256);
    size_t off[11];
    for (int i = 0; i < n; ++i) o = cn_plan(items[i], H, o, off);
    return o;
}

// Fills the ContNet table and the trainer's job list; patches the problems' step arguments to "values" mode
// (continuation values come from the problem's `cont` row).  `table_host` is the BatchProb table batch_build made.
void batch_cn_build(const BatchItem* items, int n, int hidden, const uint64_t* seeds, double lr, char* slab2,
                    void* table_host, void* cn_table_host, MlpBatchJob* jobs, int* max_cn_blocks, int64_t* max_paths)
{
    BatchProb* tab = (BatchProb*)table_host;
    CnProb* cn = (CnProb*)cn_table_host;
    const int H = cn_padded_width(hidden);
    const int np = H * 8 + H * H + H + H + 1;
    size_t o = align_up(sizeof(double) * 8 * (size_t)n, 256);
    size_t off[11];
    *max_cn_blocks = 0;
    *max_paths = 0;
    for (int i = 0; i < n; ++i) {
        const size_t end = cn_plan(items[i], H, o, off);
        CnProb& c = cn[i];
        c.cnt = (int32_t*)(slab2 + off[0]); c.s1 = (double*)(slab2 + off[1]); c.s2 = (double*)(slab2 + off[2]);
        c.offs = (int64_t*)(slab2 + off[3]);
        c.hdr = (double*)slab2 + 8 * (size_t)i;
        c.data = (float*)(slab2 + off[4]); c.cont = (float*)(slab2 + off[5]);
        c.params = (float*)(slab2 + off[6]); c.m = c.params + np; c.v = c.params + 2 * (size_t)np;
        c.wt = (float*)(slab2 + off[7]);
        c.nblk = (int)((items[i].n_paths + kCnSpan - 1) / kCnSpan);
        c.h = hidden; c.H = H; c.np = np;
        c.k0 = (uint32_t)seeds[i]; c.k1 = (uint32_t)(seeds[i] >> 32);
        tab[i].step.cont = c.cont;
        tab[i].step.ldc = 0;  // one row, rewritten every step
        MlpBatchJob& j = jobs[i];
        j = MlpBatchJob{};
        j.data = c.data; j.nrows = 0; j.batch = 0; j.first_step = 0;
        j.params = c.params; j.adam_m = c.m; j.adam_v = c.v;
        j.partial = (float*)(slab2 + off[8]); j.wt = c.wt;
        j.loss_acc = c.hdr + 5;  // (unused sum of the steps' losses)
        j.lr = lr; j.seed = 0; j.shuffle_key = 0;
        j.nrows_dev = c.hdr;
        *max_cn_blocks = std::max(*max_cn_blocks, c.nblk);
        *max_paths = std::max<int64_t>(*max_paths, items[i].n_paths);
        o = end;
    }
}

template <int H>
static hipError_t cn_forward_batch(hipStream_t st, const BatchProb* pr, const CnProb* cp, int n, int64_t max_paths, int t)
{
    constexpr size_t lds = sizeof(float) * (size_t)(H * 8 + H * H + H + H + 1);
    static std::atomic<uint64_t> attr_mask{0};
    if (lds > 48 * 1024) {
        hipError_t e = set_max_dynamic_lds(attr_mask, reinterpret_cast<const void*>(cn_forward_batch_kernel<H>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((cn_forward_batch_kernel<H>), dim3((unsigned)((max_paths + kCnBlock - 1) / kCnBlock), (unsigned)n),
                       dim3(kCnBlock), lds, st, pr, cp, t);
    return hipGetLastError();
}

// the whole backward induction of the batch: for every time step the chain count -> scan + fresh net -> rows ->
// `epochs` full-batch steps -> continuation values -> decision, each ONE launch for all problems
hipError_t batch_contnet(hipStream_t st, const void* table_dev, const void* cn_table_dev, const void* mlp_table_dev,
                         int n, const BatchExtents& e, int hidden, int epochs, int max_cn_blocks, int64_t max_paths,
                         const double* bc1_dev, const double* bc2_dev, int* tile_prefix_dev)
{
    const BatchProb* pr = (const BatchProb*)table_dev;
    const CnProb* cp = (const CnProb*)cn_table_dev;
    const int H = cn_padded_width(hidden);
    const unsigned z = (unsigned)n;
    const int Nmax = e.max_steps;
    const bool big = lsm_step_block_threads() == 1024;
    const dim3 gs((unsigned)e.sweep_blocks, 1, z), bs(big ? 1024 : 512);
    // workgroups per problem in the trainer's launches: enough to fill the chip when the batch is small, few when
    // the batch itself does (a regression set is mostly a small fraction of the paths: surplus workgroups only exit)
    // trainer launches: a lone problem gets one workgroup per possible tile; a batch shares a fixed pool of
    // workgroups over the tiles all its problems really have (work list: sizes differ wildly between problems)
    const int tiles_max = (int)((max_paths + 31) / 32);
    const bool listed = n > 1 && tile_prefix_dev != nullptr && H <= 64;  // (the 128-unit instance of the list kernel spills)
    const int gx = listed ? 4096 : tiles_max;
    auto step = [&](int t) {
        if (big) {
            if (e.vec4) hipLaunchKernelGGL((lsm_step_batch_kernel<0, 4, 1024>), gs, bs, 0, st, pr, t);
            else hipLaunchKernelGGL((lsm_step_batch_kernel<0, 1, 1024>), gs, bs, 0, st, pr, t);
        } else {
            if (e.vec4) hipLaunchKernelGGL((lsm_step_batch_kernel<0, 4, 512>), gs, bs, 0, st, pr, t);
            else hipLaunchKernelGGL((lsm_step_batch_kernel<0, 1, 512>), gs, bs, 0, st, pr, t);
        }
    };
    for (int t = Nmax; t >= 1; --t) {
        if (t < Nmax) {  // (at t = Nmax no problem has a regression step: t == N is the initialising launch)
            const dim3 gc((unsigned)max_cn_blocks, z);
            hipLaunchKernelGGL(cn_count_batch_kernel, gc, dim3(kCnBlock), 0, st, pr, cp, t);
            hipLaunchKernelGGL(cn_scan_init_batch_kernel, dim3(1, z), dim3(1024), 0, st, pr, cp, t);
            hipLaunchKernelGGL(cn_rows_batch_kernel, gc, dim3(kCnBlock), 0, st, pr, cp, t);
            if (listed) {
                hipError_t err = mlp_tile_prefix(st, mlp_table_dev, n, tile_prefix_dev);
                if (err != hipSuccess) return err;
            }
            for (int ep = 0; ep < epochs; ++ep) {
                hipError_t err = mlp_train_step_batch(st, mlp_table_dev, n, H, gx, ep, bc1_dev, bc2_dev,
                                                      listed ? tile_prefix_dev : nullptr);
                if (err != hipSuccess) return err;
            }
            hipError_t err = H == 32 ? cn_forward_batch<32>(st, pr, cp, n, max_paths, t)
                           : H == 64 ? cn_forward_batch<64>(st, pr, cp, n, max_paths, t)
                                     : cn_forward_batch<128>(st, pr, cp, n, max_paths, t);
            if (err != hipSuccess) return err;
        }
        step(t);
    }
    const dim3 gf((unsigned)e.block_blocks, 1, z);
    if (e.vec4) hipLaunchKernelGGL((lsm_final_batch_kernel<4>), gf, dim3(kBlock), 0, st, pr);
    else hipLaunchKernelGGL((lsm_final_batch_kernel<1>), gf, dim3(kBlock), 0, st, pr);
    hipLaunchKernelGGL(lsm_finalize_batch_kernel, dim3(1, 1, z), dim3(kBlock), 0, st, pr, 0);
    hipLaunchKernelGGL(cn_total_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pr, cp, n);
    return hipGetLastError();
}

}  // namespace omc
