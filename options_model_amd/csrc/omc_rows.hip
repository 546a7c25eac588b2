// omc_rows.hip -- pass 1 of the NN flow (options_model_3.py:482-563) as kernels: every in-the-money
// (step, path) of a device path matrix becomes one training row [7 normalised features, normalised
// target], in the reference's order (steps N-1 down to 1, paths ascending within a step), with the
// normalisers (feature means / population stds, target mean / std; zero std -> 1) computed on the
// way.  Nothing but the row matrix itself is materialised: ONE sweep over S counts the rows of every (step, tile) and
// forms the statistics (round 5; until then: count, sums, squared deviations = three sweeps), a scan turns the counts
// into row offsets, one more sweep writes the rows.
// The statistics are the reference's two-pass mean / population variance (:550-563) computed in one pass without its
// cancellation problem: a thread accumulates sums of d = f - c and d^2 around ITS OWN first row's values c (so a
// constant column has d = 0 exactly: variance exactly 0, and the std-0 -> 1 rule of :562 sees a true zero), turns them
// into (n, mean, M2 = sum (f - mean)^2), and the triples are merged pairwise by Chan's formula -- the one the reference
// itself uses for its streaming European pricer (:33-49) -- in a fixed tree order: bitwise reproducible, float64
// throughout, agreeing with the two-pass values to ~1e-15 relative.
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

constexpr int kTChunk = 32;  // time steps per workgroup of the three sweeps (RowsArgs::tchunk)

// A thread's spots of the chunk's time steps, ALL requested before the first is used: the loops below synchronise the
// workgroup twice per time step (the tile's count / ranks), and a load issued inside such a loop is waited for on the
// spot -- one HBM latency per time step, 32 per workgroup (round 5 profile: the count sweep ran at 2.7 TB/s, the fused
// statistics sweep at 0.8).
__device__ __forceinline__ void load_chunk(const float* col, int64_t ld, int t0, int t1, float (&sv)[kTChunk])
{
#pragma unroll
    for (int i = 0; i < kTChunk; ++i) sv[i] = t0 + i < t1 ? __builtin_nontemporal_load(col + (int64_t)(t0 + i) * ld) : 0.0f;
}

// cnt[(N-1-t) * ntiles + tile] = in-the-money paths of the tile at step t  (t = 1 .. N-1)
__global__ __launch_bounds__(kBlock) void rows_count_kernel(RowsArgs a, int32_t* __restrict__ cnt)
{
    __shared__ int wsum[kTChunk][kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t p = (int64_t)tile * kBlock + tid;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : 0);
    const int t0 = 1 + blockIdx.y * kTChunk, t1 = min(t0 + kTChunk, a.N);
    float sv[kTChunk];
    load_chunk(col, a.ld, t0, t1, sv);
#pragma unroll
    for (int i = 0; i < kTChunk; ++i) {
        const bool f = live && t0 + i < t1 && itm(sv[i], a.K, a.is_put);
        const int wc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
        if (lane == 0) wsum[i][wave] = wc;
    }
    __syncthreads();
    if (tid < t1 - t0) cnt[(size_t)(a.N - 1 - (t0 + tid)) * a.ntiles + tile] = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
}

// Exclusive prefix of counts, one workgroup of 1024 threads per SEGMENT: workgroup b scans cnt[b * seg .. b * seg + len)
// (len = min(seg, n - b * seg)) into offs[same positions] and stores the segment's total in tot[b].  A segment is walked
// in tiles of 1024 x 8 counts: a thread owns 8 CONSECUTIVE counts of the tile (a wave reads 2 KB contiguous), scans them
// in registers, the workgroup scans the 1024 thread sums through LDS and a running carry links the tiles.
// One segment (seg >= n): the flat scan, total also in offs[n].  The row sweeps use TWO levels instead -- one segment per
// time step, then one more launch over the per-step totals -- because a lone workgroup walking config 5's 980k counts
// took 1.05 ms (round 5 profile), most of pass 1.
template <class In>
__global__ __launch_bounds__(1024) void rows_scan_kernel(const In* __restrict__ cnt, int64_t n, int64_t seg,
                                                         int64_t* __restrict__ offs, int64_t* __restrict__ tot)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry_sh;
    const int tid = threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * seg, last = first + seg < n ? first + seg : n;
    int64_t carry = 0;
    for (int64_t base = first; base < last; base += 1024 * 8) {
        const int64_t lo = base + (int64_t)tid * 8;
        int64_t c[8];
        int64_t s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c[k] = lo + k < last ? (int64_t)cnt[lo + k] : 0;
            s += c[k];
        }
        part[tid] = s;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the thread sums
            const int64_t v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int64_t run = carry + (tid ? part[tid - 1] : 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (lo + k < last) offs[lo + k] = run;
            run += c[k];
        }
        if (tid == 1023) carry_sh = carry + part[1023];
        __syncthreads();
        carry = carry_sh;
        __syncthreads();
    }
    if (tid == 0) {
        if (tot) tot[blockIdx.x] = carry;
        if (seg >= n) offs[n] = carry;
    }
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

// ---- (n, mean[7], M2[7]) triples and their merge
constexpr int kTrip = 16;  // doubles per stored triple: n, mean[7], M2[7], pad
struct Trip {
    double n, mean[7], m2[7];
};

// Chan / Golub / LeVeque: b merged into a.  A side without rows leaves the other untouched.
__device__ __forceinline__ void trip_merge(Trip& a, const Trip& b)
{
    if (b.n == 0.0) return;
    if (a.n == 0.0) {
        a = b;
        return;
    }
    const double n = a.n + b.n, wb = b.n / n, wab = a.n * wb;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const double delta = b.mean[q] - a.mean[q];
        a.mean[q] += delta * wb;
        a.m2[q] += b.m2[q] + delta * delta * wab;
    }
    a.n = n;
}

// the workgroup's 256 triples -> thread 0's, by a fixed binary tree through LDS (lds: kBlock x kTrip doubles)
__device__ __forceinline__ void trip_block_merge(Trip& t, double* lds)
{
    const int tid = threadIdx.x;
    double* mine = lds + (size_t)tid * kTrip;
    mine[0] = t.n;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        mine[1 + q] = t.mean[q];
        mine[8 + q] = t.m2[q];
    }
    __syncthreads();
    for (int stride = kBlock / 2; stride >= 1; stride >>= 1) {
        if (tid < stride) {
            const double* o = lds + (size_t)(tid + stride) * kTrip;
            Trip b;
            b.n = o[0];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                b.mean[q] = o[1 + q];
                b.m2[q] = o[8 + q];
            }
            trip_merge(t, b);
            mine[0] = t.n;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                mine[1 + q] = t.mean[q];
                mine[8 + q] = t.m2[q];
            }
        }
        __syncthreads();
    }
}

// ONE sweep: cnt[(N-1-t) * ntiles + tile] as rows_count_kernel, and the workgroup's (n, mean, M2) triple of the seven
// quantities [x, x^2, x^3, max(x-1,0), s, x*s, y] over its in-the-money (step, path) pairs -> part[wg][kTrip]
__global__ __launch_bounds__(kBlock) void rows_count_stats_kernel(RowsArgs a, int32_t* __restrict__ cnt, double* __restrict__ part)
{
    __shared__ int wsum[kTChunk][kBlock / 64];
    __shared__ double lds[kBlock * kTrip];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t p = (int64_t)tile * kBlock + tid;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : 0);
    const int t0 = 1 + blockIdx.y * kTChunk, t1 = min(t0 + kTChunk, a.N);
    float sv[kTChunk];
    load_chunk(col, a.ld, t0, t1, sv);
    const double pn = payoff_d(col[(int64_t)a.N * a.ld], a.K, a.is_put);
    const double payN = pn > 0.0 ? pn : 0.0;
    double c[7], sd[7], sq[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) c[q] = sd[q] = sq[q] = 0.0;
    double n = 0.0;
#pragma unroll 4
    for (int i = 0; i < kTChunk; ++i) {
        const int t = t0 + i;
        const float s = sv[i];
        const bool f = live && t < t1 && itm(s, a.K, a.is_put);
        const int wc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
        if (lane == 0) wsum[i][wave] = wc;
        if (!f) continue;
        double v[8];
        row_values((double)s, payN, a.K, sqrt(fmax(a.T - (double)t * a.dt, 1e-6)), a.D[a.N - t], v);
        if (n == 0.0) {
#pragma unroll
            for (int q = 0; q < 7; ++q) c[q] = v[q];
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const double d = v[q] - c[q];
            sd[q] += d;
            sq[q] = __builtin_fma(d, d, sq[q]);
        }
        n += 1.0;
    }
    __syncthreads();
    if (tid < t1 - t0) cnt[(size_t)(a.N - 1 - (t0 + tid)) * a.ntiles + tile] = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
    Trip tr;
    tr.n = n;
    const double inv = n > 0.0 ? 1.0 / n : 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        tr.mean[q] = c[q] + sd[q] * inv;
        const double m2 = sq[q] - sd[q] * sd[q] * inv;
        tr.m2[q] = m2 > 0.0 ? m2 : 0.0;
    }
    trip_block_merge(tr, lds);
    if (tid == 0) {
        double* o = part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kTrip;
        o[0] = tr.n;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            o[1 + q] = tr.mean[q];
            o[8 + q] = tr.m2[q];
        }
    }
}

// part[0 .. nwg) -> out[kTrip]: thread i folds partials i, i + 256, ... in index order, then the same tree
__global__ __launch_bounds__(kBlock) void rows_merge_kernel(const double* __restrict__ part, int nwg, double* __restrict__ out)
{
    __shared__ double lds[kBlock * kTrip];
    Trip t;
    t.n = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) t.mean[q] = t.m2[q] = 0.0;
    for (int i = threadIdx.x; i < nwg; i += kBlock) {
        const double* o = part + (size_t)i * kTrip;
        Trip b;
        b.n = o[0];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            b.mean[q] = o[1 + q];
            b.m2[q] = o[8 + q];
        }
        trip_merge(t, b);
    }
    trip_block_merge(t, lds);
    if (threadIdx.x == 0) {
        out[0] = t.n;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            out[1 + q] = t.mean[q];
            out[8 + q] = t.m2[q];
        }
        out[15] = 0.0;
    }
}

struct RowsNorm {
    double fm[7], rs[7];  // feature means and reciprocal stds (feature 0 is the constant 1)
    double ym, rys;
};

// rows in the reference's order: rowbase[N-1-t] (rows of the later steps) + offs[(N-1-t) * ntiles + tile] (rows of the
// step's earlier tiles) + rank of the path among the tile's in-the-money paths at step t
__global__ __launch_bounds__(kBlock) void rows_write_kernel(RowsArgs a, RowsNorm nm, const int64_t* __restrict__ offs,
                                                            const int64_t* __restrict__ rowbase, float* __restrict__ data, int64_t cap)
{
    __shared__ int wsum[kTChunk][kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t p = (int64_t)tile * kBlock + tid;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : 0);
    const int t0 = 1 + blockIdx.y * kTChunk, t1 = min(t0 + kTChunk, a.N);
    float sv[kTChunk];
    load_chunk(col, a.ld, t0, t1, sv);
    const double pn = payoff_d(col[(int64_t)a.N * a.ld], a.K, a.is_put);
    const double payN = pn > 0.0 ? pn : 0.0;
#pragma unroll
    for (int i = 0; i < kTChunk; ++i) {
        const bool f = live && t0 + i < t1 && itm(sv[i], a.K, a.is_put);
        const int wc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
        if (lane == 0) wsum[i][wave] = wc;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < kTChunk; ++i) {
        const int t = t0 + i;
        const float s = sv[i];
        const bool f = live && t < t1 && itm(s, a.K, a.is_put);
        const uint64_t b = __builtin_amdgcn_ballot_w64(f);
        if (!f) continue;
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[i][w];
        const int rank = woff + __builtin_popcountll(b & ((1ull << lane) - 1ull));
        const int r = a.N - 1 - t;
        const int64_t row = rowbase[r] + offs[(size_t)r * a.ntiles + tile] + rank;
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

static size_t rows_nwg(int64_t M, int N, int tchunk)
{
    return (size_t)((M + kBlock - 1) / kBlock) * (size_t)((N - 1 + tchunk - 1) / tchunk > 0 ? (N - 1 + tchunk - 1) / tchunk : 1);
}

size_t nn_rows_scratch_bytes(int64_t M, int N)
{
    const size_t n = (size_t)(N - 1 > 0 ? N - 1 : 0) * (size_t)((M + kBlock - 1) / kBlock);
    return sizeof(int64_t) * (n + 2 + 2 * ((size_t)N + 2)) + sizeof(int32_t) * (n + 2) +
           sizeof(double) * (kTrip * (rows_nwg(M, N, kTChunk) + 2) + 32);
}

// scratch layout: offs int64[n+2] | rowtot int64[N+2] | rowbase int64[N+2] | cnt int32[n] | part double[nwg][kTrip] | out double[kTrip]
static void carve(void* scratch, size_t n, size_t nwg, int N, int64_t** offs, int64_t** rowtot, int64_t** rowbase, int32_t** cnt,
                  double** part, double** out)
{
    char* b = (char*)scratch;
    *offs = (int64_t*)b;
    b += sizeof(int64_t) * (n + 2);
    *rowtot = (int64_t*)b;
    b += sizeof(int64_t) * ((size_t)N + 2);
    *rowbase = (int64_t*)b;
    b += sizeof(int64_t) * ((size_t)N + 2);
    *cnt = (int32_t*)b;
    b += (sizeof(int32_t) * (n + 2) + 7) / 8 * 8;
    *part = (double*)b;
    *out = *part + kTrip * (nwg + 1);
}

static RowsArgs make_args(const LsmProblem& p, const double* D)
{
    RowsArgs a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.T = p.T; a.dt = p.T / (double)p.N; a.D = D;
    a.ntiles = (int)((p.M + kBlock - 1) / kBlock);
    a.tchunk = kTChunk;
    return a;
}

// counts + scan; *total_dev points at the device int64 holding R afterwards.  with_stats: the same sweep also forms the
// statistics, *stats_dev then points at 16 device doubles: n, mean[7], M2[7] (sum of squared deviations) of
// [x, x^2, x^3, max(x-1,0), s, x*s, y].
hipError_t nn_rows_count(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const int64_t** total_dev,
                         bool with_stats, const double** stats_dev)
{
    const RowsArgs a = make_args(p, D);
    const size_t n = (size_t)(p.N - 1) * a.ntiles, nwg = rows_nwg(p.M, p.N, a.tchunk);
    int64_t *offs, *rowtot, *rowbase; int32_t* cnt; double *part, *out;
    carve(scratch, n, nwg, p.N, &offs, &rowtot, &rowbase, &cnt, &part, &out);
    const int nrow = p.N - 1 > 0 ? p.N - 1 : 0;
    const dim3 grid(a.ntiles, (p.N - 1 + a.tchunk - 1) / a.tchunk);
    if (n > 0) {
        if (with_stats) hipLaunchKernelGGL(rows_count_stats_kernel, grid, dim3(kBlock), 0, st, a, cnt, part);
        else hipLaunchKernelGGL(rows_count_kernel, grid, dim3(kBlock), 0, st, a, cnt);
        // two-level scan: every time step's tiles in their own workgroup, then the per-step totals
        hipLaunchKernelGGL(rows_scan_kernel<int32_t>, dim3((unsigned)nrow), dim3(1024), 0, st, (const int32_t*)cnt, (int64_t)n,
                           (int64_t)a.ntiles, offs, rowtot);
    }
    hipLaunchKernelGGL(rows_scan_kernel<int64_t>, dim3(1), dim3(1024), 0, st, (const int64_t*)rowtot, (int64_t)nrow,
                       (int64_t)(nrow > 0 ? nrow : 1), rowbase, (int64_t*)nullptr);
    if (with_stats) {
        hipLaunchKernelGGL(rows_merge_kernel, dim3(1), dim3(kBlock), 0, st, part, n > 0 ? (int)(grid.x * grid.y) : 0, out);
        *stats_dev = out;
    }
    *total_dev = rowbase + nrow;
    return hipGetLastError();
}

hipError_t nn_scan_counts(hipStream_t st, const int32_t* cnt, int64_t n, int64_t* offs)
{
    hipLaunchKernelGGL(rows_scan_kernel<int32_t>, dim3(1), dim3(1024), 0, st, cnt, n, n > 0 ? n : (int64_t)1, offs, (int64_t*)nullptr);
    return hipGetLastError();
}

hipError_t nn_rows_half_counts(hipStream_t st, const LsmProblem& p, int64_t half, int64_t* counts_dev)
{
    if (p.N < 2) return hipSuccess;
    const RowsArgs a = make_args(p, nullptr);
    hipLaunchKernelGGL(rows_half_count_kernel, dim3(p.N - 1), dim3(kBlock), 0, st, a, half, counts_dev);
    return hipGetLastError();
}

hipError_t nn_rows_write(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const double* feat_mean,
                         const double* feat_std, double y_mean, double y_std, float* data, int64_t cap)
{
    const RowsArgs a = make_args(p, D);
    const size_t n = (size_t)(p.N - 1) * a.ntiles;
    if (n == 0) return hipSuccess;
    int64_t *offs, *rowtot, *rowbase; int32_t* cnt; double *part, *out;
    carve(scratch, n, rows_nwg(p.M, p.N, a.tchunk), p.N, &offs, &rowtot, &rowbase, &cnt, &part, &out);
    RowsNorm nm;
    for (int i = 0; i < 7; ++i) {
        nm.fm[i] = feat_mean[i];
        nm.rs[i] = 1.0 / feat_std[i];
    }
    nm.ym = y_mean;
    nm.rys = 1.0 / y_std;
    hipLaunchKernelGGL(rows_write_kernel, dim3(a.ntiles, (p.N - 1 + a.tchunk - 1) / a.tchunk), dim3(kBlock), 0, st, a,
                       nm, offs, rowbase, data, cap);
    return hipGetLastError();
}

}  // namespace omc
