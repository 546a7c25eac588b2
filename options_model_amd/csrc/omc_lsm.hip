// omc_lsm.hip -- Longstaff-Schwartz backward induction kernels for gfx950 (MI355X).
//
// Replaces the reference's per-step torch/numpy chains (mask -> gather -> regress ->
// compare -> scatter, with host syncs every step):
//   per-step flow   Options_model.py:108-157, options_model_2.py:278-313
//   two-pass flow   options_model_3/options_model_3.py:482-516 (pass 1), :615-651 (pass 2)
//   GPU intent      options_model_3/option_model_3_gpu.py:705-721, :804-831
// The regressor is OLS on [1,u,u^2], u = S/K - 1: eight double sums per time step, reduced
// per block through LDS, solved on chip; nothing is gathered, compacted or materialised.
//
// Per-path state is (sx, tex) = spot and step index of the path's current exercise time
// (tex == N: terminal payoff).  A cash-flow seen from step t is payoff(sx) * D[tex - t]
// with D[k] = exp(-r dt k): no per-step rescaling pass, no rounding drift, and the state is
// only rewritten when a path actually exercises.
//
// All kernels: 256-thread workgroups, 16-byte loads per lane where alignment allows,
// fixed-order reductions (bitwise reproducible), no atomics.
#include "omc_lsm_dev.h"

#include <cstdlib>
#include <cstring>

namespace omc {

// ------------------------------------------------------------------ __global__ entry points
template <int SEM, int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lsm_step_kernel(StepArgs a)
{
    lsm_step_body<SEM, VEC, BLOCK>(a, blockIdx.x, a.nblk);
}

// the same with the argument block in device memory: the N launches of one sweep differ only in `t`,
// so a captured HIP graph of them can be replayed for any pricing of the same geometry after
// refreshing that block
template <int SEM, int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lsm_step_ind_kernel(const StepArgs* __restrict__ ap, int t)
{
    StepArgs a = *ap;
    a.t = t;
    lsm_step_body<SEM, VEC, BLOCK>(a, blockIdx.x, a.nblk);
}

__global__ __launch_bounds__(kBlock) void lsm_reduce_step_kernel(const double* part, double* gmom, int t,
                                                                 int nblk, int pstride, int gstride)
{
    lsm_reduce_step_body(part, gmom, t, nblk, pstride, gstride);
}

template <int VEC, int TPW, int PUT>
__global__ __launch_bounds__(kBlock) void lsm_pass1_kernel(Pass1Args a) { lsm_pass1_body<VEC, TPW, PUT>(a); }

__global__ __launch_bounds__(kBlock) void lsm_reduce_pass1_kernel(const double* part1, double* gmom,
                                                                  int64_t ntiles, int N)
{
    lsm_reduce_pass1_body(part1, gmom, ntiles, N);
}

template <int VEC, bool WRITE_STATE>
__global__ __launch_bounds__(kBlock) void lsm_pass2_kernel(Pass2Args a) { lsm_pass2_body<VEC, WRITE_STATE>(a); }

// the two sweeps on the antithetic-folded matrix (omc_lsm_dev.h)
template <int VEC, int TPW, int PUT>
__global__ __launch_bounds__(kBlock) void lsm_pass1_fold_kernel(Pass1Args a) { lsm_pass1_fold_body<VEC, TPW, PUT>(a); }
template <int VEC, int PUT>
__global__ __launch_bounds__(kBlock) void lsm_pass2_fold_kernel(Pass2Args a) { lsm_pass2_fold_body<VEC, PUT>(a); }

// cK[t] = c0 g^t, t = 0 .. N, by N sequential float64 multiplications (IEEE: the host oracle repeats them exactly)
__global__ void lsm_fold_table_kernel(double* __restrict__ cK, int N, double c0, double g)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double c = c0;
    cK[0] = c;
    for (int t = 1; t <= N; ++t) {
        c *= g;
        cK[t] = c;
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void lsm_final_kernel(FinalArgs a) { lsm_final_body<VEC>(a); }

template <int VEC>
__global__ __launch_bounds__(kBlock) void lsm_final_ind_kernel(const FinalArgs* __restrict__ ap)
{
    lsm_final_body<VEC>(*ap);
}

__global__ __launch_bounds__(kBlock) void lsm_finalize_kernel(const double* part, const double* gmom,
                                                              double* result, int nblk, int N, int pstride)
{
    lsm_finalize_body(part, gmom, result, nblk, N, pstride);
}

struct FinalizeArgs {
    const double* part;
    const double* gmom;
    double* result;
    int nblk, N, pstride, gstride;
};
// device-resident argument block of a captured per-step sweep
struct SweepArgs {
    StepArgs step;
    FinalArgs fin;
    FinalizeArgs fz;
};

__global__ __launch_bounds__(kBlock) void lsm_finalize_ind_kernel(const FinalizeArgs* __restrict__ ap)
{
    const FinalizeArgs a = *ap;
    lsm_finalize_body(a.part, a.gmom, a.result, a.nblk, a.N, a.pstride, a.gstride);
}

// ---- K pricings of one geometry advanced by ONE launch per time step (omc_price_american_seq, per-step flows).
// `tab[k]` holds pricing k's arguments; the chip's workgroups are dealt G per pricing (pricing = blockIdx.x / G),
// so all K pricings are resident together and workgroup w of a pricing walks the slots w, w + G, ... of the
// single-pricing geometry (lsm_step_body): per pricing the bits of its own launch, per launch K x the bytes.
template <int SEM, int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lsm_step_multi_kernel(const SweepArgs* __restrict__ tab, int G, int t)
{
    const int k = (int)blockIdx.x / G;
    StepArgs a = tab[k].step;
    a.t = t;
    lsm_step_body<SEM, VEC, BLOCK>(a, (int)blockIdx.x - k * G, G);
}

__global__ __launch_bounds__(kBlock) void lsm_reduce_step_multi_kernel(const SweepArgs* __restrict__ tab, int t)
{
    const StepArgs& a = tab[blockIdx.x].step;
    lsm_reduce_step_body(a.part, a.gmom, t, a.nblk, a.pstride, a.gstride);
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void lsm_final_multi_kernel(const SweepArgs* __restrict__ tab)
{
    lsm_final_body<VEC>(tab[blockIdx.z].fin);
}

__global__ __launch_bounds__(kBlock) void lsm_finalize_multi_kernel(const SweepArgs* __restrict__ tab)
{
    const FinalizeArgs a = tab[blockIdx.x].fz;
    lsm_finalize_body(a.part, a.gmom, a.result, a.nblk, a.N, a.pstride, a.gstride);
}

// ------------------------------------------------------------------ host launchers
static inline bool vec4_ok(const LsmProblem& p)
{
    return (p.M % 4) == 0 && (p.ld % 4) == 0 && ((uintptr_t)p.S % 16) == 0;
}

int lsm_step_blocks(int64_t M)
{
    const int64_t per_block = (int64_t)kBlock * 4;
    int64_t b = (M + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > kMaxLsmBlocks ? kMaxLsmBlocks : b);
}

int lsm_step_block_threads()
{
    // threads per workgroup of the per-step sweep: 1024 (one workgroup per CU, 256 partials) or 512
    static const int v = [] {
        const char* e = getenv("OMC_STEP_BLOCK");
        const int x = e ? atoi(e) : 0;
        return x == 512 ? 512 : 1024;
    }();
    return v;
}

int lsm_sweep_blocks(int64_t M)
{
    const int64_t per_block = (int64_t)lsm_step_block_threads() * 4;
    int64_t b = (M + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > kStepMaxBlocks ? kStepMaxBlocks : b);
}

size_t lsm_part1_tiles(int64_t M)
{
    // sized for the VEC=1 fallback too (4x more tiles); the kernel uses what it needs
    return (size_t)((M + kBlock - 1) / kBlock);
}

static void fill_step_args(StepArgs& a, const LsmProblem& p, const LsmWorkspace& w, int t, bool external_moments)
{
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.invK = 1.0 / p.K;
    a.sx = w.sx; a.tex = w.tex; a.live = w.live; a.D = w.D; a.part = w.part; a.gmom = w.gmom; a.betas = w.betas;
    a.t = t; a.nblk = lsm_sweep_blocks(p.M); a.external = external_moments ? 1 : 0;
    a.pstride = kPStride;
    a.gstride = w.gstride;
    a.cont = w.cont; a.ldc = w.ldc;
}

template <int SEM, int VEC, int BLOCK>
static void launch_step(hipStream_t st, const StepArgs& a, const StepArgs* ind, int t, size_t dyn)
{
    const dim3 grid(a.nblk), block(BLOCK);
    if (ind) hipLaunchKernelGGL((lsm_step_ind_kernel<SEM, VEC, BLOCK>), grid, block, dyn, st, ind, t);
    else hipLaunchKernelGGL((lsm_step_kernel<SEM, VEC, BLOCK>), grid, block, dyn, st, a);
}

// `ind` != null: the kernel reads its arguments from that device block (see lsm_step_args)
static hipError_t lsm_step_impl(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int semantics,
                                int t, bool external_moments, const StepArgs* ind)
{
    StepArgs a;
    fill_step_args(a, p, w, t, external_moments);
    const bool v4 = vec4_ok(p);
    const size_t dyn = semantics == 1 ? sizeof(double) * (size_t)(p.N + 1) : 0;
    const bool big = lsm_step_block_threads() == 1024;
#define OMC_STEP(SEM, VEC)                                                       \
    do {                                                                         \
        if (big) launch_step<SEM, VEC, 1024>(st, a, ind, t, dyn);               \
        else launch_step<SEM, VEC, 512>(st, a, ind, t, dyn);                    \
    } while (0)
    if (semantics == 0) {
        if (v4) OMC_STEP(0, 4); else OMC_STEP(0, 1);
    } else {
        if (v4) OMC_STEP(1, 4); else OMC_STEP(1, 1);
    }
#undef OMC_STEP
    return hipGetLastError();
}

hipError_t lsm_step(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int semantics,
                    int t, bool external_moments)
{
    return lsm_step_impl(st, p, w, semantics, t, external_moments, nullptr);
}

size_t lsm_sweep_args_bytes() { return sizeof(SweepArgs); }

static void fill_final_args(FinalArgs& a, const LsmProblem& p, const LsmWorkspace& w, int tval, bool use_flags,
                            bool fill_state)
{
    a.sx = w.sx; a.tex = w.tex; a.live = use_flags ? w.live : nullptr;
    a.M = p.M; a.N = p.N; a.is_put = p.is_put; a.tval = tval; a.fill_state = fill_state ? 1 : 0;
    a.K = p.K; a.D = w.D; a.part = w.part;
    a.nblk = lsm_step_blocks(p.M); a.pstride = kPStride;
}

// host image of the device argument block the indirect kernels read
void lsm_sweep_args_image(const LsmProblem& p, const LsmWorkspace& w, int semantics, bool fill_state, void* out,
                          bool external_moments)
{
    SweepArgs s;
    memset(&s, 0, sizeof s);
    fill_step_args(s.step, p, w, 0, external_moments);
    fill_final_args(s.fin, p, w, semantics == 1 ? 0 : 1, semantics == 0, fill_state);
    s.fz.part = w.part; s.fz.gmom = w.gmom; s.fz.result = w.result;
    s.fz.nblk = s.fin.nblk; s.fz.N = p.N; s.fz.pstride = kPStride; s.fz.gstride = w.gstride;
    memcpy(out, &s, sizeof s);
}

// workgroups per pricing when K pricings share the launches of a per-step sweep: the chip's workgroups (one
// 1024-thread workgroup per CU) divided by K, at least what keeps a workgroup within kStepMaxItems slots
int lsm_multi_groups(int64_t M, int K, int device_cus)
{
    const int nblk = lsm_sweep_blocks(M);
    const int wgs = device_cus > 0 ? device_cus : 256;
    int G = wgs / (K > 0 ? K : 1);
    const int gmin = (nblk + kStepMaxItems - 1) / kStepMaxItems;
    if (G < gmin) G = gmin;
    if (G < 1) G = 1;
    return G > nblk ? nblk : G;
}

hipError_t lsm_step_multi(hipStream_t st, const void* table_dev, int K, int G, int semantics, bool vec4, int N, int t)
{
    const SweepArgs* tab = (const SweepArgs*)table_dev;
    const dim3 grid((unsigned)(K * G));
    const size_t dyn = semantics == 1 ? sizeof(double) * (size_t)(N + 1) : 0;
    const bool big = lsm_step_block_threads() == 1024;
#define OMC_STEPM(SEM, VEC)                                                                                    \
    do {                                                                                                       \
        if (big) hipLaunchKernelGGL((lsm_step_multi_kernel<SEM, VEC, 1024>), grid, dim3(1024), dyn, st, tab, G, t); \
        else hipLaunchKernelGGL((lsm_step_multi_kernel<SEM, VEC, 512>), grid, dim3(512), dyn, st, tab, G, t);  \
    } while (0)
    if (semantics == 0) {
        if (vec4) OMC_STEPM(0, 4); else OMC_STEPM(0, 1);
    } else {
        if (vec4) OMC_STEPM(1, 4); else OMC_STEPM(1, 1);
    }
#undef OMC_STEPM
    return hipGetLastError();
}

hipError_t lsm_reduce_step_moments_multi(hipStream_t st, const void* table_dev, int K, int t)
{
    hipLaunchKernelGGL(lsm_reduce_step_multi_kernel, dim3(K), dim3(kBlock), 0, st, (const SweepArgs*)table_dev, t);
    return hipGetLastError();
}

// valuation + finalize of all K pricings: two launches
hipError_t lsm_final_multi(hipStream_t st, const void* table_dev, int K, int64_t M)
{
    const SweepArgs* tab = (const SweepArgs*)table_dev;
    const int nblk = lsm_step_blocks(M);
    if ((M % 4) == 0) hipLaunchKernelGGL((lsm_final_multi_kernel<4>), dim3(nblk, 1, K), dim3(kBlock), 0, st, tab);
    else hipLaunchKernelGGL((lsm_final_multi_kernel<1>), dim3(nblk, 1, K), dim3(kBlock), 0, st, tab);
    hipLaunchKernelGGL(lsm_finalize_multi_kernel, dim3(K), dim3(kBlock), 0, st, tab);
    return hipGetLastError();
}

// the whole per-step sweep (N step launches + valuation + finalize) with device-resident arguments:
// this is what gets captured into a HIP graph.  Launch geometry depends on (M, N, semantics, ld, S
// alignment) only -- the graph's cache key.
hipError_t lsm_sweep_indirect(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int semantics,
                              const void* args_dev)
{
    const SweepArgs* sa = (const SweepArgs*)args_dev;
    for (int t = p.N; t >= 1; --t) {
        hipError_t e = lsm_step_impl(st, p, w, semantics, t, false, &sa->step);
        if (e != hipSuccess) return e;
    }
    const int nblk = lsm_step_blocks(p.M);
    if ((p.M % 4) == 0) hipLaunchKernelGGL((lsm_final_ind_kernel<4>), dim3(nblk), dim3(kBlock), 0, st, &sa->fin);
    else hipLaunchKernelGGL((lsm_final_ind_kernel<1>), dim3(nblk), dim3(kBlock), 0, st, &sa->fin);
    hipLaunchKernelGGL(lsm_finalize_ind_kernel, dim3(1), dim3(kBlock), 0, st, &sa->fz);
    return hipGetLastError();
}

hipError_t lsm_reduce_step_moments(hipStream_t st, const LsmWorkspace& w, int t, int nblk)
{
    hipLaunchKernelGGL(lsm_reduce_step_kernel, dim3(1), dim3(kBlock), 0, st, w.part, w.gmom, t, nblk,
                       kPStride, w.gstride);
    return hipGetLastError();
}

hipError_t lsm_fold_table(hipStream_t st, double* cK, int N, double c0, double g)
{
    hipLaunchKernelGGL(lsm_fold_table_kernel, dim3(1), dim3(64), 0, st, cK, N, c0, g);
    return hipGetLastError();
}

// pass 1 on the folded matrix: P = M / 2 stored columns, two tiles of 64 x VEC columns per wave and step (= 1,024 paths, as
// in the full sweep; OMC_FOLD_TPW = 1 | 4 for experiments)
static hipError_t lsm_pass1_moments_fold(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w)
{
    Pass1Args a;
    const int64_t P = p.M / 2;
    a.S = p.S; a.ld = p.ld; a.M = P; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.invK = 1.0 / p.K; a.D = w.D; a.part1 = w.part1; a.cK = p.fold_cK;
    const bool v4 = (P % 4) == 0 && (p.ld % 4) == 0 && ((uintptr_t)p.S % 16) == 0;
    static const int tpw_env = getenv("OMC_FOLD_TPW") ? atoi(getenv("OMC_FOLD_TPW")) : 0;
    const int tpw = (v4 && (tpw_env == 1 || tpw_env == 4)) ? tpw_env : 2;
    const int64_t per_wave = 64 * (int64_t)(v4 ? 4 : 1) * tpw;
    a.ntiles = (P + per_wave - 1) / per_wave;
    // Steps per workgroup: the folded sweep is bound by its float64 arithmetic, not by the rows it reads, so what counts is
    // that every CU stays busy to the end -- many short workgroups (measured at C2, 245 tile-workgroups: chunks of 16-32
    // steps 0.145-0.147 ms, 63 steps 0.156, 84 steps -- one resident round -- 0.162, 126 steps 0.183; 8M paths: 32).
    static const int tch_env = getenv("OMC_PASS1_TCHUNK") ? atoi(getenv("OMC_PASS1_TCHUNK")) : 0;
    const int64_t wgs_x = (a.ntiles + 3) / 4;
    a.tchunk = (tch_env >= 2 && tch_env <= kFoldMaxChunk) ? tch_env : 32;
    const dim3 grid((unsigned)wgs_x, (unsigned)((p.N - 1 + a.tchunk - 1) / a.tchunk));
    if (w.ev_p1_begin) (void)hipEventRecord(w.ev_p1_begin, st);
    if (v4 && tpw == 1) {
        if (p.is_put) hipLaunchKernelGGL((lsm_pass1_fold_kernel<4, 1, 1>), grid, dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((lsm_pass1_fold_kernel<4, 1, 0>), grid, dim3(kBlock), 0, st, a);
    } else if (v4 && tpw == 4) {
        if (p.is_put) hipLaunchKernelGGL((lsm_pass1_fold_kernel<4, 4, 1>), grid, dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((lsm_pass1_fold_kernel<4, 4, 0>), grid, dim3(kBlock), 0, st, a);
    } else if (v4) {
        if (p.is_put) hipLaunchKernelGGL((lsm_pass1_fold_kernel<4, 2, 1>), grid, dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((lsm_pass1_fold_kernel<4, 2, 0>), grid, dim3(kBlock), 0, st, a);
    } else {
        if (p.is_put) hipLaunchKernelGGL((lsm_pass1_fold_kernel<1, 2, 1>), grid, dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((lsm_pass1_fold_kernel<1, 2, 0>), grid, dim3(kBlock), 0, st, a);
    }
    if (w.ev_p1_end) (void)hipEventRecord(w.ev_p1_end, st);
    hipLaunchKernelGGL(lsm_reduce_pass1_kernel, dim3(p.N - 1), dim3(kBlock), 0, st, w.part1, w.gmom, a.ntiles, p.N);
    return hipGetLastError();
}

hipError_t lsm_pass1_moments(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w)
{
    if (p.N < 2) return hipSuccess;
    if (p.fold_cK) return lsm_pass1_moments_fold(st, p, w);
    Pass1Args a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.invK = 1.0 / p.K; a.D = w.D; a.part1 = w.part1;
    const bool v4 = vec4_ok(p);
    // tuning knobs (defaults measured on MI355X): tiles per wave, steps per block
    static const int tpw_env = getenv("OMC_PASS1_TPW") ? atoi(getenv("OMC_PASS1_TPW")) : 0;
    static const int tch_env = getenv("OMC_PASS1_TCHUNK") ? atoi(getenv("OMC_PASS1_TCHUNK")) : 0;
    const int tpw = (v4 && (tpw_env == 1 || tpw_env == 2 || tpw_env == 8)) ? tpw_env : 4;
    const int64_t per_wave = 64 * (int64_t)(v4 ? 4 : 1) * tpw;  // paths per wave per step
    a.ntiles = (p.M + per_wave - 1) / per_wave;
    // Steps per workgroup.  The kernel holds 3 waves per SIMD (152 VGPRs), i.e. 3 workgroups per CU: when all the
    // workgroups of a launch fit on the chip at once there is no partly filled last round of dispatch (measured at
    // C2, 245 tile-workgroups: 8 chunks of 32 steps = 2.55 rounds 0.214 ms, 4 x 63 = 1.28 rounds 0.230, 3 x 84 =
    // 0.96 round 0.207), so take the most chunks that still fit; larger problems run many rounds and keep 32.
    // The partition changes no sum: partials are per (step, tile).
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n > 0 ? n : 256;
    }();
    int tchunk = 32;
    const int64_t wgs_x = (a.ntiles + 3) / 4;
    if (wgs_x <= 3 * (int64_t)cus && p.N > 2) {
        const int chunks = (int)((3 * (int64_t)cus) / wgs_x);
        const int t = (p.N - 1 + chunks - 1) / chunks;
        if (t <= 126 && t >= 32) tchunk = t;
    }
    a.tchunk = (tch_env >= 2 && tch_env <= kPass1MaxChunk) ? tch_env : tchunk;
    const dim3 grid((unsigned)((a.ntiles + 3) / 4), (unsigned)((p.N - 1 + a.tchunk - 1) / a.tchunk));
    if (w.ev_p1_begin) (void)hipEventRecord(w.ev_p1_begin, st);
    auto launch = [&](auto vec, auto tp) {
        constexpr int V = decltype(vec)::value, T = decltype(tp)::value;
        if (p.is_put) hipLaunchKernelGGL((lsm_pass1_kernel<V, T, 1>), grid, dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((lsm_pass1_kernel<V, T, 0>), grid, dim3(kBlock), 0, st, a);
    };
    using std::integral_constant;
    if (!v4) launch(integral_constant<int, 1>{}, integral_constant<int, 4>{});
    else if (tpw == 1) launch(integral_constant<int, 4>{}, integral_constant<int, 1>{});
    else if (tpw == 2) launch(integral_constant<int, 4>{}, integral_constant<int, 2>{});
    else if (tpw == 8) launch(integral_constant<int, 4>{}, integral_constant<int, 8>{});
    else launch(integral_constant<int, 4>{}, integral_constant<int, 4>{});
    if (w.ev_p1_end) (void)hipEventRecord(w.ev_p1_end, st);
    hipLaunchKernelGGL(lsm_reduce_pass1_kernel, dim3(p.N - 1), dim3(kBlock), 0, st, w.part1, w.gmom,
                       a.ntiles, p.N);
    return hipGetLastError();
}

hipError_t lsm_pass2_apply(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w,
                           bool write_state, bool solve_from_moments)
{
    Pass2Args a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.invK = 1.0 / p.K; a.D = w.D; a.betas = w.betas; a.sx = w.sx; a.tex = w.tex;
    a.part = w.part;
    if (solve_from_moments) {
        a.gmom = w.gmom;
        a.betas_out = w.betas;
    }
    static const int fvec_env = getenv("OMC_FOLD_P2_VEC") ? atoi(getenv("OMC_FOLD_P2_VEC")) : 0;
    // columns per thread of the folded pass 2: 2 (8-byte loads, twice the threads) until the 16-byte form alone fills the
    // chip with workgroups (measured: C2's 0.5M columns 0.131 against 0.136 ms, C3's 4M columns 1.026 against 1.008)
    const int fvec = (fvec_env == 1 || fvec_env == 2 || fvec_env == 4) ? fvec_env : ((p.M / 2) >= (int64_t(1) << 21) ? 4 : 2);
    const int nblk = lsm_step_blocks(p.fold_cK ? (p.M / 2) * (4 / fvec) : p.M);
    a.nblk = nblk; a.pstride = kPStride;
    const size_t dyn = sizeof(double) * 4 * (size_t)(p.N + 1);
    const bool v4 = vec4_ok(p);
    if (w.ev_p2_begin) (void)hipEventRecord(w.ev_p2_begin, st);
    if (p.fold_cK) {  // the folded matrix: M / 2 stored columns, both partners decided from every spot; no state arrays
        if (write_state) return hipErrorInvalidValue;
        a.M = p.M / 2;
        a.cK = p.fold_cK;
        const bool f4 = (a.M % 4) == 0 && (p.ld % 4) == 0 && ((uintptr_t)p.S % 16) == 0;
        auto launch = [&](auto vec) {
            constexpr int V = decltype(vec)::value;
            if (p.is_put) hipLaunchKernelGGL((lsm_pass2_fold_kernel<V, 1>), dim3(nblk), dim3(kBlock), dyn, st, a);
            else hipLaunchKernelGGL((lsm_pass2_fold_kernel<V, 0>), dim3(nblk), dim3(kBlock), dyn, st, a);
        };
        if (f4 && fvec == 4) launch(std::integral_constant<int, 4>{});
        else if (f4 && fvec == 2) launch(std::integral_constant<int, 2>{});
        else launch(std::integral_constant<int, 1>{});
    } else if (v4) {
        if (write_state) hipLaunchKernelGGL((lsm_pass2_kernel<4, true>), dim3(nblk), dim3(kBlock), dyn, st, a);
        else hipLaunchKernelGGL((lsm_pass2_kernel<4, false>), dim3(nblk), dim3(kBlock), dyn, st, a);
    } else {
        if (write_state) hipLaunchKernelGGL((lsm_pass2_kernel<1, true>), dim3(nblk), dim3(kBlock), dyn, st, a);
        else hipLaunchKernelGGL((lsm_pass2_kernel<1, false>), dim3(nblk), dim3(kBlock), dyn, st, a);
    }
    if (w.ev_p2_end) (void)hipEventRecord(w.ev_p2_end, st);
    hipLaunchKernelGGL(lsm_finalize_kernel, dim3(1), dim3(kBlock), 0, st, w.part, w.gmom, w.result,
                       nblk, p.N, kPStride);
    return hipGetLastError();
}

hipError_t lsm_finalize(hipStream_t st, const double* part, const double* gmom, double* result,
                        int nblk, int N)
{
    hipLaunchKernelGGL(lsm_finalize_kernel, dim3(1), dim3(kBlock), 0, st, part, gmom, result, nblk, N,
                       kPStride);
    return hipGetLastError();
}

hipError_t lsm_final_reduce(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int tval, bool use_flags,
                            bool fill_state)
{
    FinalArgs a;
    fill_final_args(a, p, w, tval, use_flags, fill_state);
    const int nblk = a.nblk;
    if ((p.M % 4) == 0) hipLaunchKernelGGL((lsm_final_kernel<4>), dim3(nblk), dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL((lsm_final_kernel<1>), dim3(nblk), dim3(kBlock), 0, st, a);
    hipLaunchKernelGGL(lsm_finalize_kernel, dim3(1), dim3(kBlock), 0, st, w.part, w.gmom, w.result,
                       nblk, p.N, kPStride);
    return hipGetLastError();
}

}  // namespace omc
