// omc_rows.hip -- pass 1 of the NN flow (options_model_3.py:482-563) as kernels: every in-the-money
// (step, path) of a device path matrix becomes one training row [7 normalised features, normalised
// target], in the reference's order (steps N-1 down to 1, paths ascending within a step), with the
// normalisers (feature means / population stds, target mean / std; zero std -> 1) computed on the
// way.  Nothing but the row matrix itself is materialised: counts -> scan -> two statistics passes
// over S -> one write pass, each a streaming read of S.
#include "omc_device.h"
#include "omc_kernels.h"

namespace omc {

namespace {

struct RowsArgs {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, T, dt;
    const double* D;   // D[k] = exp(-r dt k)
    int ntiles;        // tiles of 256 paths
    int tchunk;        // time steps per workgroup
};

__device__ __forceinline__ bool itm(float s, double K, int is_put) { return payoff_d(s, K, is_put) > 0.0; }

// cnt[(N-1-t) * ntiles + tile] = in-the-money paths of the tile at step t  (t = 1 .. N-1)
__global__ __launch_bounds__(kBlock) void rows_count_kernel(RowsArgs a, int32_t* __restrict__ cnt)
{
    __shared__ int wsum[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t p = (int64_t)tile * kBlock + tid;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : 0);
    const int t0 = 1 + blockIdx.y * a.tchunk, t1 = min(t0 + a.tchunk, a.N);
    for (int t = t0; t < t1; ++t) {
        const bool f = live && itm(col[(int64_t)t * a.ld], a.K, a.is_put);
        const int wc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
        if (lane == 0) wsum[wave] = wc;
        __syncthreads();
        if (tid == 0) cnt[(size_t)(a.N - 1 - t) * a.ntiles + tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// exclusive prefix of cnt[0..n) into offs[0..n), total into offs[n]; one workgroup of 1024 threads.
// The array is walked in tiles of 1024 x 8 counts: a thread owns 8 CONSECUTIVE counts of the tile (a wave reads 2 KB
// contiguous), scans them in registers, the workgroup scans the 1024 thread sums through LDS and a running carry links
// the tiles.  (Until round 4 every thread walked its own contiguous 1/1024 of the array: two passes of fully
// uncoalesced loads, 2.6 ms for config 5's 980k counts; integer sums, so the result is the same.)
__global__ __launch_bounds__(1024) void rows_scan_kernel(const int32_t* __restrict__ cnt, int64_t n,
                                                         int64_t* __restrict__ offs)
{
    __shared__ int64_t seg[1024];
    __shared__ int64_t carry_sh;
    const int tid = threadIdx.x;
    int64_t carry = 0;
    for (int64_t base = 0; base < n; base += 1024 * 8) {
        const int64_t lo = base + (int64_t)tid * 8;
        int c[8];
        int64_t s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c[k] = lo + k < n ? cnt[lo + k] : 0;
            s += c[k];
        }
        seg[tid] = s;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the thread sums
            const int64_t v = tid >= d ? seg[tid - d] : 0;
            __syncthreads();
            seg[tid] += v;
            __syncthreads();
        }
        int64_t run = carry + (tid ? seg[tid - 1] : 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (lo + k < n) offs[lo + k] = run;
            run += c[k];
        }
        if (tid == 1023) carry_sh = carry + seg[1023];
        __syncthreads();
        carry = carry_sh;
        __syncthreads();
    }
    if (tid == 0) offs[n] = carry;
}

// half[(N-1-t) * 2 + h] = in-the-money paths at step t among columns [0, half) (h = 0) / [half, M) (h = 1);
// one workgroup per time step
__global__ __launch_bounds__(kBlock) void rows_half_count_kernel(RowsArgs a, int64_t half, int64_t* __restrict__ out)
{
    __shared__ long long acc[2][kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = 1 + (int)blockIdx.x;
    if (t >= a.N) return;
    const float* rowp = a.S + (int64_t)t * a.ld;
    long long c0 = 0, c1 = 0;
    for (int64_t p0 = 0; p0 < a.M; p0 += kBlock) {
        const int64_t p = p0 + tid;
        const bool f = p < a.M && itm(rowp[p], a.K, a.is_put);
        const uint64_t b0 = __builtin_amdgcn_ballot_w64(f && p < half), b1 = __builtin_amdgcn_ballot_w64(f && p >= half);
        c0 += __builtin_popcountll(b0);
        c1 += __builtin_popcountll(b1);
    }
    if (lane == 0) {
        acc[0][wave] = c0;
        acc[1][wave] = c1;
    }
    __syncthreads();
    if (tid < 2) {
        long long s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += acc[tid][w];
        out[(size_t)(a.N - 1 - t) * 2 + tid] = s;
    }
}

// features of one in-the-money (t, path): [x, x^2, x^3, max(x-1,0), s, x*s] and the target y
__device__ __forceinline__ void row_values(double sd, double payN, double K, double st, double disc, double (&f)[8])
{
    const double x = sd / K;
    f[0] = x;
    f[1] = x * x;
    f[2] = x * x * x;
    f[3] = fmax(x - 1.0, 0.0);
    f[4] = st;
    f[5] = x * st;
    f[6] = payN * disc;
    f[7] = 1.0;
}

// PASS 0: sums of the 7 quantities (+ count in slot 7); PASS 1: squared deviations from `mean`
template <int PASS>
__global__ __launch_bounds__(kBlock) void rows_stats_kernel(RowsArgs a, const double* __restrict__ mean,
                                                            double* __restrict__ part, int pstride)
{
    __shared__ double red[kNQ * kRedStride];
    extern __shared__ double sst[];  // s_t = sqrt(max(T - t dt, 1e-6)), t = 0 .. N
    for (int t = threadIdx.x; t <= a.N; t += kBlock) sst[t] = sqrt(fmax(a.T - (double)t * a.dt, 1e-6));
    __syncthreads();
    double acc[8], mu[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        acc[q] = 0.0;
        mu[q] = PASS ? mean[q] : 0.0;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < a.M; p += stride) {
        const float* col = a.S + p;
        const double pn = payoff_d(col[(int64_t)a.N * a.ld], a.K, a.is_put);
        const double payN = pn > 0.0 ? pn : 0.0;
        for (int t = a.N - 1; t >= 1; --t) {
            const float s = col[(int64_t)t * a.ld];
            if (!itm(s, a.K, a.is_put)) continue;
            double f[8];
            row_values((double)s, payN, a.K, sst[t], a.D[a.N - t], f);
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const double d = f[q] - mu[q];
                acc[q] += PASS ? d * d : d;
            }
            acc[7] += 1.0;
        }
    }
    const double r = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0) part[(size_t)(threadIdx.x >> 3) * pstride + blockIdx.x] = r;
}

// part[q][0..nblk) summed in index order -> out[q]
__global__ __launch_bounds__(kBlock) void rows_finish_kernel(const double* part, int nblk, int pstride, double* out)
{
    __shared__ double red[kNQ * kRedStride];
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += part[(size_t)q * pstride + i];
    }
    const double r = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0) out[threadIdx.x >> 3] = r;
}

struct RowsNorm {
    double fm[7], rs[7];  // feature means and reciprocal stds (feature 0 is the constant 1)
    double ym, rys;
};

// rows in the reference's order: offs[(N-1-t) * ntiles + tile] + rank of the path among the tile's
// in-the-money paths at step t
__global__ __launch_bounds__(kBlock) void rows_write_kernel(RowsArgs a, RowsNorm nm, const int64_t* __restrict__ offs,
                                                            float* __restrict__ data, int64_t cap)
{
    __shared__ int wsum[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t p = (int64_t)tile * kBlock + tid;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : 0);
    const double pn = payoff_d(col[(int64_t)a.N * a.ld], a.K, a.is_put);
    const double payN = pn > 0.0 ? pn : 0.0;
    const int t0 = 1 + blockIdx.y * a.tchunk, t1 = min(t0 + a.tchunk, a.N);
    for (int t = t0; t < t1; ++t) {
        const float s = col[(int64_t)t * a.ld];
        const bool f = live && itm(s, a.K, a.is_put);
        const uint64_t b = __builtin_amdgcn_ballot_w64(f);
        if (lane == 0) wsum[wave] = __builtin_popcountll(b);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int rank = woff + __builtin_popcountll(b & ((1ull << lane) - 1ull));
        __syncthreads();
        if (!f) continue;
        const int64_t row = offs[(size_t)(a.N - 1 - t) * a.ntiles + tile] + rank;
        if (row >= cap) continue;
        double v[8];
        row_values((double)s, payN, a.K, sqrt(fmax(a.T - (double)t * a.dt, 1e-6)), a.D[a.N - t], v);
        float4 lo, hi;
        lo.x = (float)((1.0 - nm.fm[0]) * nm.rs[0]);
        lo.y = (float)((v[0] - nm.fm[1]) * nm.rs[1]);
        lo.z = (float)((v[1] - nm.fm[2]) * nm.rs[2]);
        lo.w = (float)((v[2] - nm.fm[3]) * nm.rs[3]);
        hi.x = (float)((v[3] - nm.fm[4]) * nm.rs[4]);
        hi.y = (float)((v[4] - nm.fm[5]) * nm.rs[5]);
        hi.z = (float)((v[5] - nm.fm[6]) * nm.rs[6]);
        hi.w = (float)((v[6] - nm.ym) * nm.rys);
        float4* dst = reinterpret_cast<float4*>(data + row * 8);
        dst[0] = lo;
        dst[1] = hi;
    }
}

}  // namespace

size_t nn_rows_scratch_bytes(int64_t M, int N)
{
    const size_t n = (size_t)(N - 1 > 0 ? N - 1 : 0) * (size_t)((M + kBlock - 1) / kBlock);
    return sizeof(int64_t) * (n + 2) + sizeof(int32_t) * (n + 2) + sizeof(double) * (8 * 1024 + 32);
}

// scratch layout: offs int64[n+1] | cnt int32[n] | part double[8][1024] | out double[16]
static void carve(void* scratch, size_t n, int64_t** offs, int32_t** cnt, double** part, double** out)
{
    char* b = (char*)scratch;
    *offs = (int64_t*)b;
    b += sizeof(int64_t) * (n + 2);
    *cnt = (int32_t*)b;
    b += (sizeof(int32_t) * (n + 2) + 7) / 8 * 8;
    *part = (double*)b;
    *out = *part + 8 * 1024;
}

static RowsArgs make_args(const LsmProblem& p, const double* D)
{
    RowsArgs a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.T = p.T; a.dt = p.T / (double)p.N; a.D = D;
    a.ntiles = (int)((p.M + kBlock - 1) / kBlock);
    a.tchunk = 32;
    return a;
}

// counts + scan; *total_dev points at the device int64 holding R afterwards
hipError_t nn_rows_count(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const int64_t** total_dev)
{
    const RowsArgs a = make_args(p, D);
    const size_t n = (size_t)(p.N - 1) * a.ntiles;
    int64_t* offs; int32_t* cnt; double *part, *out;
    carve(scratch, n, &offs, &cnt, &part, &out);
    if (n > 0)
        hipLaunchKernelGGL(rows_count_kernel, dim3(a.ntiles, (p.N - 1 + a.tchunk - 1) / a.tchunk), dim3(kBlock), 0, st,
                           a, cnt);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, (int64_t)n, offs);
    *total_dev = offs + n;
    return hipGetLastError();
}

hipError_t nn_scan_counts(hipStream_t st, const int32_t* cnt, int64_t n, int64_t* offs)
{
    hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, n, offs);
    return hipGetLastError();
}

hipError_t nn_rows_half_counts(hipStream_t st, const LsmProblem& p, int64_t half, int64_t* counts_dev)
{
    if (p.N < 2) return hipSuccess;
    const RowsArgs a = make_args(p, nullptr);
    hipLaunchKernelGGL(rows_half_count_kernel, dim3(p.N - 1), dim3(kBlock), 0, st, a, half, counts_dev);
    return hipGetLastError();
}

// PASS 0 -> sums_host[0..7] (slot 7 = row count); PASS 1 (mean_host[0..7]) -> squared deviations.
// Synchronises the stream (the 8 results go back to the host).
hipError_t nn_rows_stats(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, int pass,
                         const double* mean_host, double* sums_host)
{
    const RowsArgs a = make_args(p, D);
    const size_t n = (size_t)(p.N - 1) * a.ntiles;
    int64_t* offs; int32_t* cnt; double *part, *out;
    carve(scratch, n, &offs, &cnt, &part, &out);
    int nblk = (int)((p.M + kBlock - 1) / kBlock);
    nblk = nblk > 1024 ? 1024 : nblk;
    const size_t dyn = sizeof(double) * (size_t)(p.N + 1);
    hipError_t e;
    if (pass == 0) {
        hipLaunchKernelGGL(rows_stats_kernel<0>, dim3(nblk), dim3(kBlock), dyn, st, a, (const double*)nullptr, part, 1024);
    } else {
        if ((e = hipMemcpyAsync(out + 16, mean_host, sizeof(double) * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(rows_stats_kernel<1>, dim3(nblk), dim3(kBlock), dyn, st, a, (const double*)(out + 16), part, 1024);
    }
    hipLaunchKernelGGL(rows_finish_kernel, dim3(1), dim3(kBlock), 0, st, part, nblk, 1024, out);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(sums_host, out, sizeof(double) * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

hipError_t nn_rows_write(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const double* feat_mean,
                         const double* feat_std, double y_mean, double y_std, float* data, int64_t cap)
{
    const RowsArgs a = make_args(p, D);
    const size_t n = (size_t)(p.N - 1) * a.ntiles;
    if (n == 0) return hipSuccess;
    int64_t* offs; int32_t* cnt; double *part, *out;
    carve(scratch, n, &offs, &cnt, &part, &out);
    RowsNorm nm;
    for (int i = 0; i < 7; ++i) {
        nm.fm[i] = feat_mean[i];
        nm.rs[i] = 1.0 / feat_std[i];
    }
    nm.ym = y_mean;
    nm.rys = 1.0 / y_std;
    hipLaunchKernelGGL(rows_write_kernel, dim3(a.ntiles, (p.N - 1 + a.tchunk - 1) / a.tchunk), dim3(kBlock), 0, st, a,
                       nm, offs, data, cap);
    return hipGetLastError();
}

}  // namespace omc
