// omc_ols7.hip -- the v3 two-pass flow (options_model_3.py:482-516 pass 1, :542-563 normalisation, :615-651 pass 2) with
// ONE global least-squares fit on the reference's seven features (create_regression_features, :105-121) in place of
// the network: regressor "ols7", between the per-step 3-term polynomial and the NN regressor.  The reference has no such
// regressor (lsm_poly_degree is validated and ignored); the flow is the reference's, the fit is OLS on its
// own normalised design matrix -- what tools/capture_golden.py records as `ols7_*` from the reference's own
// create_regression_features, and what oracle/reference_flow.py two_pass_ols7_regressor restates.
//
// Pass 1 never materialises the R x 7 design matrix (R ~ 0.46 N M rows): ONE sweep over the path matrix forms, per
// lane, the sums of d_i and d_i d_j around the first row it sees (d = g - c: a constant column has d = 0 exactly) of
// g = [u, u^2, u^3, max(u, 0), s, u s, y], u = x - 1 (the reference's features in a better conditioned basis),
// turns them into (n, mean[7], C[28] = sum (f_i - mean_i)(f_j - mean_j)) and merges those pairwise by Chan's formula
// in a fixed tree (bitwise reproducible, float64).  The host builds the normal equations of the STANDARDISED columns
// from C -- the correlation matrix: the same fit as lstsq on (X - mean) / std, zero-variance columns (the constant,
// max(x - 1, 0) for a put) dropped = lstsq's minimum-norm weight 0 -- and solves 6 x 6.  Pass 2: one thread per path
// walks backward with the fitted weights, float64, strict >, sticky.
#include "omc_device.h"
#include "omc_kernels.h"

namespace omc {

namespace {

constexpr int kCoBlock = 128;  // threads per workgroup of the co-moment sweep
constexpr int kCoChunk = 32;   // time steps requested at a time (32 loads in flight per thread)
constexpr int kCoIter = 4;     // such chunks per workgroup: one block merge (7 levels of 40 doubles through LDS) per 128 steps
constexpr int kCoFold = 512;   // partials folded by one workgroup of the first merge stage
constexpr int kCoGroup = 8;    // steps whose in-the-money rows a wave packs before it works on them
static_assert(kCoChunk * kCoIter <= kCoBlock, "one thread per step fills the workgroup's per-step tables");
constexpr int kCoQ = 7;        // u, u^2, u^3, max(u,0), s, u*s, y with u = x - 1
constexpr int kCoC = 28;       // upper triangle of the 7 x 7 co-moment matrix
constexpr int kCo = 40;        // doubles per stored triple: n, mean[7], C[28], pad

__host__ __device__ constexpr int tri(int i, int j) { return i * kCoQ - i * (i - 1) / 2 + (j - i); }  // i <= j

struct CoTrip {
    double n, mean[kCoQ], c[kCoC];
};

__device__ __forceinline__ void co_merge(CoTrip& a, const CoTrip& b)
{
    if (b.n == 0.0) return;
    if (a.n == 0.0) {
        a = b;
        return;
    }
    const double n = a.n + b.n, wb = b.n / n, wab = a.n * wb;
    double delta[kCoQ];
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) delta[q] = b.mean[q] - a.mean[q];
#pragma unroll
    for (int i = 0; i < kCoQ; ++i)
#pragma unroll
        for (int j = i; j < kCoQ; ++j) a.c[tri(i, j)] += b.c[tri(i, j)] + delta[i] * delta[j] * wab;
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) a.mean[q] += delta[q] * wb;
    a.n = n;
}

__device__ __forceinline__ void co_store(double* o, const CoTrip& t)
{
    o[0] = t.n;
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) o[1 + q] = t.mean[q];
#pragma unroll
    for (int q = 0; q < kCoC; ++q) o[1 + kCoQ + q] = t.c[q];
}

__device__ __forceinline__ void co_load(const double* o, CoTrip& t)
{
    t.n = o[0];
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) t.mean[q] = o[1 + q];
#pragma unroll
    for (int q = 0; q < kCoC; ++q) t.c[q] = o[1 + kCoQ + q];
}

// the workgroup's triples -> thread 0's, by a fixed binary tree (thread i takes thread i + stride's) through LDS: at every
// level only the upper half hands its triples over, so the buffer holds kCoBlock / 2 of them (20 KB: LDS no longer limits
// the sweep to three workgroups per CU)
__device__ __forceinline__ void co_block_merge(CoTrip& t, double* lds)
{
    const int tid = threadIdx.x;
    for (int stride = kCoBlock / 2; stride >= 1; stride >>= 1) {
        __syncthreads();  // (the buffer is free: the previous level's readers -- or the sweep's queues -- are done)
        if (tid >= stride && tid < 2 * stride) co_store(lds + (size_t)(tid - stride) * kCo, t);
        __syncthreads();
        if (tid < stride) {
            CoTrip b;
            co_load(lds + (size_t)tid * kCo, b);
            co_merge(t, b);
        }
    }
}

struct CoArgs {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, T, dt;
    double inv_K;     // x = S * (1 / K) in this sweep (the moments; pass 2 divides, like the oracle)
    const double* D;  // D[k] = exp(-r dt k)
};

// the seven quantities of one in-the-money (t, path): the reference's features (options_model_3.py:105-121, columns 1..6;
// column 0 is the constant) re-expressed in u = x - 1 -- the same span, a better conditioned Gram matrix; the host states
// means, stds and weights for the reference's own features -- and the pass-1 target :491-516 (terminal payoff discounted to t)
__device__ __forceinline__ void co_values(double sd, double payN, double inv_K, double st, double disc, double (&f)[kCoQ])
{
    const double u = sd * inv_K - 1.0;  // x - 1: the powers of a number near 0 instead of a number near 1 (see omc_lsm_ols7)
    f[0] = u;
    f[1] = u * u;
    f[2] = u * u * u;
    f[3] = fmax(u, 0.0);
    f[4] = st;
    f[5] = u * st;
    f[6] = payN * disc;
}

__global__ __launch_bounds__(kCoBlock) void ols7_comoment_kernel(CoArgs a, double* __restrict__ part)
{
    __shared__ double lds[kCoBlock / 2 * kCo];
    __shared__ double s_st[kCoChunk * kCoIter], s_disc[kCoChunk * kCoIter];  // per step: sqrt(max(T - t dt, 1e-6)), D[N - t]
    const int tid = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * kCoBlock + tid;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : 0);
    const int tb = 1 + blockIdx.y * (kCoChunk * kCoIter), te = min(tb + kCoChunk * kCoIter, a.N);  // this workgroup's steps
    if (tb + tid < te) {  // (what every thread would otherwise recompute per row: a double square root)
        s_st[tid] = sqrt(fmax(a.T - (double)(tb + tid) * a.dt, 1e-6));
        s_disc[tid] = a.D[a.N - (tb + tid)];
    }
    const double pn = payoff_d(col[(int64_t)a.N * a.ld], a.K, a.is_put);
    const double payN = pn > 0.0 ? pn : 0.0;
    double c[kCoQ], sd[kCoQ], sq[kCoC];
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) c[q] = sd[q] = 0.0;
#pragma unroll
    for (int q = 0; q < kCoC; ++q) sq[q] = 0.0;
    double n = 0.0;
    __syncthreads();
    // In-the-money cells are about half of all cells and scattered over the lanes: a wave that walks its 64 paths step by
    // step runs the ~50 float64 instructions of a row for every step at which ANY lane is in the money -- all of them.
    // So the rows of kCoGroup steps are first packed into a wave-private queue (ballot + prefix count: row k of the group
    // goes to lane k % 64) and the wave makes ceil(rows / 64) passes instead of kCoGroup: 0.97 -> ?? ms at 1M x 252.
    // A lane's sums are then over rows of several paths; its shift point c is still the first row IT sees, so a constant
    // column still has d = 0 and variance 0 exactly.  (The queue lives in the block-merge buffer, used after the sweep.)
    const int lane = tid & 63;
    double* const queue = lds + (size_t)(tid >> 6) * (32 * kCo);  // per wave: kCoGroup * 64 entries of 2 doubles
    static_assert(kCoGroup * 64 * 2 <= 32 * kCo, "the queue of a wave fits its share of the merge buffer");
    float nx[kCoChunk];  // the NEXT chunk's spots: requested before the current chunk's arithmetic starts
#pragma unroll
    for (int i = 0; i < kCoChunk; ++i) nx[i] = tb + i < te ? __builtin_nontemporal_load(col + (int64_t)(tb + i) * a.ld) : 0.0f;
    for (int t0 = tb; t0 < te; t0 += kCoChunk) {
        const int t1 = min(t0 + kCoChunk, te);
        float sv[kCoChunk];
#pragma unroll
        for (int i = 0; i < kCoChunk; ++i) {
            sv[i] = nx[i];
            nx[i] = t1 + i < te ? __builtin_nontemporal_load(col + (int64_t)(t1 + i) * a.ld) : 0.0f;
        }
#pragma unroll
        for (int g = 0; g < kCoChunk / kCoGroup; ++g) {
            int rows = 0;
#pragma unroll
            for (int i = 0; i < kCoGroup; ++i) {
                const int t = t0 + g * kCoGroup + i;
                const float s = sv[g * kCoGroup + i];
                const bool itm = live && t < t1 && payoff_d(s, a.K, a.is_put) > 0.0;
                const uint64_t m = __ballot(itm);
                if (itm) {
                    const int k = rows + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    queue[2 * k] = __builtin_bit_cast(double, ((uint64_t)(uint32_t)(t - tb) << 32) | (uint64_t)__float_as_uint(s));
                    queue[2 * k + 1] = payN;
                }
                rows += __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < rows; k += 64) {
                const uint64_t e = __builtin_bit_cast(uint64_t, queue[2 * k]);
                const int ts = (int)(e >> 32);
                double v[kCoQ];
                co_values((double)__uint_as_float((uint32_t)e), queue[2 * k + 1], a.inv_K, s_st[ts], s_disc[ts], v);
                if (n == 0.0) {
#pragma unroll
                    for (int q = 0; q < kCoQ; ++q) c[q] = v[q];
                }
                double d[kCoQ];
#pragma unroll
                for (int q = 0; q < kCoQ; ++q) {
                    d[q] = v[q] - c[q];
                    sd[q] += d[q];
                }
#pragma unroll
                for (int ii = 0; ii < kCoQ; ++ii)
#pragma unroll
                    for (int jj = ii; jj < kCoQ; ++jj) sq[tri(ii, jj)] = __builtin_fma(d[ii], d[jj], sq[tri(ii, jj)]);
                n += 1.0;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    CoTrip tr;
    tr.n = n;
    const double inv = n > 0.0 ? 1.0 / n : 0.0;
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) tr.mean[q] = c[q] + sd[q] * inv;
#pragma unroll
    for (int ii = 0; ii < kCoQ; ++ii)
#pragma unroll
        for (int jj = ii; jj < kCoQ; ++jj) {
            const double v = sq[tri(ii, jj)] - sd[ii] * sd[jj] * inv;
            tr.c[tri(ii, jj)] = (ii == jj && v < 0.0) ? 0.0 : v;
        }
    co_block_merge(tr, lds);
    if (tid == 0) co_store(part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kCo, tr);
}

// Workgroup b folds part[b * per .. min((b + 1) * per, n)) into out[b]: thread i takes partials i, i + 128, ... of its slice in
// index order, then the same tree.  Two launches (per = kCoFold, then everything that is left) instead of one workgroup
// walking all of the sweep's partials: 0.52 ms -> microseconds at 1M x 252.
__global__ __launch_bounds__(kCoBlock) void ols7_merge_kernel(const double* __restrict__ part, int n, int per, double* __restrict__ out)
{
    __shared__ double lds[kCoBlock / 2 * kCo];
    CoTrip t;
    t.n = 0.0;
#pragma unroll
    for (int q = 0; q < kCoQ; ++q) t.mean[q] = 0.0;
#pragma unroll
    for (int q = 0; q < kCoC; ++q) t.c[q] = 0.0;
    const int lo = blockIdx.x * per, hi = min(lo + per, n);
    for (int i = lo + threadIdx.x; i < hi; i += kCoBlock) {
        CoTrip b;
        co_load(part + (size_t)i * kCo, b);
        co_merge(t, b);
    }
    co_block_merge(t, lds);
    if (threadIdx.x == 0) co_store(out + (size_t)blockIdx.x * kCo, t);
}

struct Ols7Apply {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, T, dt;
    double fm[7], rs[7], w[7];  // feature means, reciprocal stds, weights (column 0 = the constant)
    double ym, ysd;
    float* sx;
    int32_t* tex;
};

// pass 2 (:615-651): path p walked from t = N-1 down to 1; exercise where payoff > continuation (strict), once (sticky
// mask: the walk ends there); continuation = ((f - mean) / std) . w * Y_std + Y_mean, the oracle's expression.  Rows are
// fetched kAhead steps ahead of their use.
__global__ __launch_bounds__(256) void ols7_pass2_kernel(Ols7Apply a)
{
    constexpr int kAhead = 8, kTab = 1024;
    __shared__ double s_st[kTab];  // sqrt(max(T - t dt, 1e-6)) of the first kTab steps (beyond: per row, as before)
    for (int t = threadIdx.x; t < kTab && t <= a.N; t += 256) s_st[t] = sqrt(fmax(a.T - (double)t * a.dt, 1e-6));
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.M) return;
    const float* col = a.S + p;
    float sx = col[(int64_t)a.N * a.ld];
    int tex = a.N;
    float ring[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) ring[i] = a.N - 1 - i >= 1 ? __builtin_nontemporal_load(col + (int64_t)(a.N - 1 - i) * a.ld) : 0.0f;
    for (int t0 = a.N - 1; t0 >= 1; t0 -= kAhead) {
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            const int t = t0 - i;
            const float sf = ring[i];
            const int tn = t - kAhead;
            ring[i] = tn >= 1 ? __builtin_nontemporal_load(col + (int64_t)tn * a.ld) : 0.0f;
            if (t < 1 || tex != a.N) continue;
            const double sd = (double)sf;
            const double imm = a.is_put ? a.K - sd : sd - a.K;
            if (!(imm > 0.0)) continue;
            const double x = sd / a.K, st = t < kTab ? s_st[t] : sqrt(fmax(a.T - (double)t * a.dt, 1e-6));
            const double f[7] = {1.0, x, x * x, x * x * x, fmax(x - 1.0, 0.0), st, x * st};
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 7; ++q) acc += ((f[q] - a.fm[q]) * a.rs[q]) * a.w[q];
            const double cont = acc * a.ysd + a.ym;
            if (imm > cont) {
                tex = t;
                sx = sf;
            }
        }
        if (tex != a.N) break;  // (the thread's path has exercised: nothing left to decide)
    }
    a.sx[p] = sx;
    a.tex[p] = tex;
}

}  // namespace

size_t ols7_scratch_bytes(int64_t M, int N)
{
    const size_t per = (size_t)kCoChunk * kCoIter;
    const size_t nwg = (size_t)((M + kCoBlock - 1) / kCoBlock) * (size_t)(N - 1 > 0 ? (N - 1 + per - 1) / per : 1);
    return sizeof(double) * kCo * (nwg + (nwg + kCoFold - 1) / kCoFold + 2);
}

// -> *stats_dev: kOls7Stats doubles on the device: n, mean[7], C[28] (upper triangle, row-major) of
// [u, u^2, u^3, max(u,0), s, u*s, y], u = x - 1, over the in-the-money (step, path) pairs
hipError_t ols7_comoments(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const double** stats_dev)
{
    CoArgs a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.T = p.T; a.dt = p.T / (double)p.N; a.D = D;
    a.inv_K = 1.0 / p.K;
    constexpr int per = kCoChunk * kCoIter;
    const int ngroup = p.N - 1 > 0 ? (p.N - 1 + per - 1) / per : 0;
    const dim3 grid((unsigned)((p.M + kCoBlock - 1) / kCoBlock), (unsigned)(ngroup > 0 ? ngroup : 1));
    double* part = (double*)scratch;
    const int nwg = ngroup > 0 ? (int)(grid.x * grid.y) : 0;
    const int nfold = (nwg + kCoFold - 1) / kCoFold;
    double* mid = part + (size_t)kCo * (size_t)(grid.x * grid.y);
    double* out = mid + (size_t)kCo * (size_t)nfold;
    if (nwg > 0) {
        hipLaunchKernelGGL(ols7_comoment_kernel, grid, dim3(kCoBlock), 0, st, a, part);
        hipLaunchKernelGGL(ols7_merge_kernel, dim3((unsigned)nfold), dim3(kCoBlock), 0, st, part, nwg, kCoFold, mid);
    }
    hipLaunchKernelGGL(ols7_merge_kernel, dim3(1), dim3(kCoBlock), 0, st, mid, nfold, nfold > 0 ? nfold : 1, out);
    *stats_dev = out;
    return hipGetLastError();
}

hipError_t ols7_pass2(hipStream_t st, const LsmProblem& p, const double* feat_mean, const double* feat_std,
                      const double* w7, double y_mean, double y_std, float* sx, int32_t* tex)
{
    Ols7Apply a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.T = p.T; a.dt = p.T / (double)p.N;
    for (int q = 0; q < 7; ++q) {
        a.fm[q] = feat_mean[q];
        a.rs[q] = 1.0 / feat_std[q];
        a.w[q] = w7[q];
    }
    a.ym = y_mean; a.ysd = y_std;
    a.sx = sx; a.tex = tex;
    hipLaunchKernelGGL(ols7_pass2_kernel, dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace omc
