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
#include "omc_lsm_dev.h"  // itm_threshold: the float32 in-the-money test of the polynomial pass 1

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

// ONE sweep: cnt[(N-1-t) * ntiles + tile] = in-the-money paths of the tile at step t (t = 1 .. N-1), and the workgroup's (n, mean, M2) triple of the seven
// quantities [x, x^2, x^3, max(x-1,0), s, x*s, y] over its in-the-money (step, path) pairs -> part[wg][kTrip].
//
// Round 6: the sweep was float64-issue-bound at 0.15 of the HBM roofline (0.84 ms at config 5: per lane and step a
// float64 division, a float64 square root, 21 operations of per-feature deviation sums behind a divergent branch, then
// 8 levels of Chan merges per workgroup).  Now, branch-free, per lane and step:
//   * in the money <=> a float32 compare against the threshold of the polynomial pass 1 (exactly payoff > 0 in float64),
//     the count on the scalar unit (popcount of the compare mask);
//   * sqrt(tau) and the discount factor of the step from a table in LDS (32 entries per workgroup);
//   * POWER SUMS around the workgroup's FIRST ROW (x_c, s_c, y_c, max(x_c - 1, 0): uniform in the workgroup) -- sum u .. u^6
//     of u = x - x_c, and with e = s_t - s_c (uniform per step): sum e, e^2, u e, u^2 e, u e^2, u^2 e^2 --, sums and squares
//     of y - y_c and of max(x-1, 0) - its centre: 17 accumulators, 26 fused operations.  Centres shared by the workgroup
//     let the lanes' sums be ADDED (three fixed-order block reductions) instead of merged; one thread turns the
//     workgroup's sums into (n, mean, M2) of the seven quantities by the binomial expansions below.
// A constant column still has variance exactly 0 where the reference's rule needs it (:562): the centre is one of the
// rows, so rows that agree in a quantity have deviation 0 in it throughout and every sum involved is an exact 0.  Float64
// throughout; deterministic (fixed summation trees); agrees with the two-pass statistics to ~1e-14 relative (the sums are
// centred inside the workgroup's own data, so the expansions cancel at most a digit).
constexpr int kNS = 17;  // n U1 U2 U3 U4 U5 U6 UE U2E UE2 U2E2 ME ME2 Y Y2 MX MX2
__global__ __launch_bounds__(kBlock) void rows_count_stats_kernel(RowsArgs a, int32_t* __restrict__ cnt, double* __restrict__ part)
{
    __shared__ int wsum[kTChunk][kBlock / 64];
    __shared__ double red[kNQ * kRedStride];
    __shared__ double tab[2][kTChunk];  // s_t = sqrt(tau_t), discount factor
    __shared__ double tot[24];
    __shared__ double centre[4];        // x, s, y, max(x - 1, 0) of the workgroup's FIRST row
    __shared__ unsigned ckey;
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
    const double invK = 1.0 / a.K;
    if (tid < kTChunk) {
        const int t = min(t0 + tid, a.N - 1);
        tab[0][tid] = sqrt(fmax(a.T - (double)t * a.dt, 1e-6));
        tab[1][tid] = a.D[a.N - t];
        if (tid == 0) ckey = 0xFFFFFFFFu;
    }
    const float thr = live ? itm_threshold(a.K, a.is_put) : (a.is_put ? -__builtin_inff() : __builtin_inff());
    const bool is_put = a.is_put != 0;
    // which of the lane's 32 steps are in the money; the workgroup's centre = its FIRST row (lowest thread that has one, at
    // that thread's first step): a real row, so that rows which are constant in a quantity -- one row in all, one time step,
    // identical spots or targets -- have deviation 0 EXACTLY and the variance comes out as a true zero (the reference's
    // "zero std -> 1" rule, :562; a centre outside the data left rounding noise of 1e-16 there: soak r06)
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < kTChunk; ++i) bits |= (uint32_t)((is_put ? sv[i] < thr : sv[i] > thr) && t0 + i < t1) << i;
    __syncthreads();
    {
        const uint64_t has = __builtin_amdgcn_ballot_w64(bits != 0u);
        if (has != 0ull && lane == (int)__builtin_ctzll(has)) atomicMin(&ckey, ((unsigned)tid << 5) | (unsigned)__builtin_ctz(bits));
    }
    __syncthreads();
    const unsigned ck = ckey;
    if (ck != 0xFFFFFFFFu && tid == (int)(ck >> 5)) {
        const int i0 = (int)(ck & 31u);
        float sc32 = sv[0];
#pragma unroll
        for (int i = 1; i < kTChunk; ++i) sc32 = i == i0 ? sv[i] : sc32;
        const double xc = (double)sc32 * invK;
        centre[0] = xc;
        centre[1] = tab[0][i0];
        centre[2] = payN * tab[1][i0];
        centre[3] = fmax(fma((double)sc32, invK, -1.0), 0.0);
    } else if (ck == 0xFFFFFFFFu && tid == 0) {
        centre[0] = centre[1] = centre[2] = centre[3] = 0.0;  // no row in this workgroup: every sum below stays 0
    }
    __syncthreads();
    const double c = centre[0], sc = centre[1], yc = centre[2], cmx = centre[3];
    double acc[kNS];
#pragma unroll
    for (int q = 0; q < kNS; ++q) acc[q] = 0.0;
#pragma unroll 4
    for (int i = 0; i < kTChunk; ++i) {
        const float s = sv[i];
        const bool f = (bits >> i) & 1u;
        const int wc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
        if (lane == 0) wsum[i][wave] = wc;
        const double m = f ? 1.0 : 0.0;          // (masking by multiplication: u m and y m are exact)
        const double e = tab[0][i] - sc, g = tab[1][i];
        const double x1 = fma((double)s, invK, -1.0);
        const double u = fma((double)s, invK, -c) * m;
        const double mx = (fmax(x1, 0.0) - cmx) * m;
        const double y = fma(payN, g, -yc) * m;
        const double u2 = u * u, u3 = u2 * u, ue = u * e, me = m * e;
        acc[0] += m;
        acc[1] += u;
        acc[2] += u2;
        acc[3] += u3;
        acc[4] = fma(u2, u2, acc[4]);
        acc[5] = fma(u3, u2, acc[5]);
        acc[6] = fma(u3, u3, acc[6]);
        acc[7] += ue;
        acc[8] = fma(u2, e, acc[8]);
        acc[9] = fma(ue, e, acc[9]);
        acc[10] = fma(ue, ue, acc[10]);
        acc[11] += me;
        acc[12] = fma(me, e, acc[12]);
        acc[13] += y;
        acc[14] = fma(y, y, acc[14]);
        acc[15] += mx;
        acc[16] = fma(mx, mx, acc[16]);
    }
    __syncthreads();
    if (tid < t1 - t0) cnt[(size_t)(a.N - 1 - (t0 + tid)) * a.ntiles + tile] = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
    // the workgroup's sums: three fixed-order reductions of eight quantities each
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double a8[kNQ];
#pragma unroll
        for (int q = 0; q < kNQ; ++q) a8[q] = 8 * r + q < kNS ? acc[8 * r + q] : 0.0;
        const double sum = block_reduce8(a8, red);
        if (tid < 64 && (tid & 7) == 0) tot[8 * r + (tid >> 3)] = sum;
        __syncthreads();
    }
    if (tid != 0) return;
    double* o = part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kTrip;
    const double n = tot[0];
    if (!(n > 0.0)) {
#pragma unroll
        for (int q = 0; q < kTrip; ++q) o[q] = 0.0;
        return;
    }
    const double U1 = tot[1], U2 = tot[2], U3 = tot[3], U4 = tot[4], U5 = tot[5], U6 = tot[6], UE = tot[7], U2E = tot[8],
                 UE2 = tot[9], U2E2 = tot[10], ME = tot[11], ME2 = tot[12], Y = tot[13], Y2 = tot[14], MX = tot[15], MX2 = tot[16];
    const double inv = 1.0 / n, c2 = c * c, c3 = c2 * c, c4 = c2 * c2;
    double sd[7], sq[7], base[7];  // sum of deviations from `base`, sum of their squares
    base[0] = c;       sd[0] = U1;                              sq[0] = U2;
    base[1] = c2;      sd[1] = 2.0 * c * U1 + U2;               sq[1] = 4.0 * c2 * U2 + 4.0 * c * U3 + U4;
    base[2] = c3;      sd[2] = 3.0 * c2 * U1 + 3.0 * c * U2 + U3;
    sq[2] = 9.0 * c4 * U2 + 18.0 * c3 * U3 + 15.0 * c2 * U4 + 6.0 * c * U5 + U6;
    base[3] = cmx;     sd[3] = MX;                              sq[3] = MX2;
    base[4] = sc;      sd[4] = ME;                              sq[4] = ME2;
    base[5] = c * sc;  sd[5] = c * ME + sc * U1 + UE;
    sq[5] = c2 * ME2 + sc * sc * U2 + U2E2 + 2.0 * c * sc * UE + 2.0 * c * UE2 + 2.0 * sc * U2E;
    base[6] = yc;      sd[6] = Y;                               sq[6] = Y2;
    o[0] = n;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        o[1 + q] = base[q] + sd[q] * inv;
        const double m2 = sq[q] - sd[q] * sd[q] * inv;
        o[8 + q] = m2 > 0.0 ? m2 : 0.0;
    }
    o[15] = 0.0;
}

// Workgroup b folds partials [b * per, min((b + 1) * per, nwg)) into out[b][kTrip]: thread i takes partials i, i + 256, ...
// of the slice in index order, then the block tree.  Two levels (round 6): config 5's 31k workgroup partials through ONE
// workgroup were 122 dependent Chan merges per thread (0.12 ms); slices of 256 -- one partial per thread -- then one
// workgroup over the slice results.  The tree is fixed by (nwg, per): bitwise reproducible.
__global__ __launch_bounds__(kBlock) void rows_merge_kernel(const double* __restrict__ part, int nwg, int per, double* __restrict__ out)
{
    __shared__ double lds[kBlock * kTrip];
    Trip t;
    t.n = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) t.mean[q] = t.m2[q] = 0.0;
    const int lo = (int)blockIdx.x * per, hi = lo + per < nwg ? lo + per : nwg;
    for (int i = lo + (int)threadIdx.x; i < hi; i += kBlock) {
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
        double* dst = out + (size_t)blockIdx.x * kTrip;
        dst[0] = t.n;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            dst[1 + q] = t.mean[q];
            dst[8 + q] = t.m2[q];
        }
        dst[15] = 0.0;
    }
}

struct RowsNorm {
    double rs[7], nb[7];  // (f - mean) / std as fma(f, rs, nb): rs = 1 / std, nb = -mean / std (feature 0 is the constant 1)
    double rys, nby;
};

// rows in the reference's order: rowbase[N-1-t] (rows of the later steps) + offs[(N-1-t) * ntiles + tile] (rows of the
// step's earlier tiles) + rank of the path among the tile's in-the-money paths at step t.  Round 6: sqrt(tau) and the
// discount factor of a step from a table in LDS (they were a float64 square root and a load per lane and step), x = S / K
// as a multiplication by 1 / K, the in-the-money test as pass 1's float32 compare, (f - mean) / std as one fma.
__global__ __launch_bounds__(kBlock) void rows_write_kernel(RowsArgs a, RowsNorm nm, const int64_t* __restrict__ offs,
                                                            const int64_t* __restrict__ rowbase, float* __restrict__ data, int64_t cap)
{
    __shared__ int wsum[kTChunk][kBlock / 64];
    __shared__ double tab[2][kTChunk];
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
    const double invK = 1.0 / a.K;
    const float thr = live ? itm_threshold(a.K, a.is_put) : (a.is_put ? -__builtin_inff() : __builtin_inff());
    const bool is_put = a.is_put != 0;
    if (tid < kTChunk) {
        const int t = min(t0 + tid, a.N - 1);
        tab[0][tid] = sqrt(fmax(a.T - (double)t * a.dt, 1e-6));
        tab[1][tid] = a.D[a.N - t];
    }
#pragma unroll
    for (int i = 0; i < kTChunk; ++i) {
        const bool f = (is_put ? sv[i] < thr : sv[i] > thr) && t0 + i < t1;
        const int wc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
        if (lane == 0) wsum[i][wave] = wc;
    }
    __syncthreads();
    // A row is a 32-byte record and the rows of a wave's in-the-money lanes are consecutive in `data` (rank order = lane
    // order).  Stored straight from the lanes that own them -- two 16-byte stores per lane with a 32-byte stride -- every
    // store instruction touches twice the lines it fills, and the kernel sat in its store queue (round 6 counters: waves
    // waiting to ISSUE 68 % of their life at 31 % VALU, 1.42 ms for 4.7 GB).  Now the records go through a wave-private LDS
    // patch in rank order and leave as CONTIGUOUS 16-byte chunks: chunk c of the wave's run by lane c, then c + 64.
    // Two patches, alternating by step: one wave barrier per step orders write -> read, the other patch's reads of the
    // previous step are behind this step's barrier.
    __shared__ float4 stage[2][kBlock / 64][128];
    const float c0 = (float)fma(1.0, nm.rs[0], nm.nb[0]);
#pragma unroll 2
    for (int i = 0; i < kTChunk; ++i) {
        const int t = t0 + i;
        if (t >= t1) break;  // (uniform)
        const float s = sv[i];
        const bool f = is_put ? s < thr : s > thr;
        const uint64_t b = __builtin_amdgcn_ballot_w64(f);
        const int cnt = __builtin_popcountll(b);
        if (cnt == 0) continue;  // (uniform per wave)
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[i][w];
        const int r = a.N - 1 - t;
        const int64_t row0 = rowbase[r] + offs[(size_t)r * a.ntiles + tile] + woff;  // first row of this wave's run
        float4* patch = stage[i & 1][wave];
        if (f) {
            const int rank = __builtin_popcountll(b & ((1ull << lane) - 1ull));
            const double st = tab[0][i], x = (double)s * invK, x2 = x * x;
            float4 lo, hi;
            lo.x = c0;
            lo.y = (float)fma(x, nm.rs[1], nm.nb[1]);
            lo.z = (float)fma(x2, nm.rs[2], nm.nb[2]);
            lo.w = (float)fma(x2 * x, nm.rs[3], nm.nb[3]);
            hi.x = (float)fma(fmax(x - 1.0, 0.0), nm.rs[4], nm.nb[4]);
            hi.y = (float)fma(st, nm.rs[5], nm.nb[5]);
            hi.z = (float)fma(x * st, nm.rs[6], nm.nb[6]);
            hi.w = (float)fma(payN * tab[1][i], nm.rys, nm.nby);
            patch[2 * rank] = lo;
            patch[2 * rank + 1] = hi;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t room = cap - row0;  // rows of this run that fit the caller's buffer
        const int nchunk = 2 * (int)(room >= cnt ? cnt : (room > 0 ? room : 0));
        float4* dst = reinterpret_cast<float4*>(data + row0 * 8);
        if (lane < nchunk) dst[lane] = patch[lane];
        if (lane + 64 < nchunk) dst[lane + 64] = patch[lane + 64];
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
           sizeof(double) * (kTrip * (rows_nwg(M, N, kTChunk) + rows_nwg(M, N, kTChunk) / kBlock + 4) + 32);
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

// counts + scan + statistics, ONE sweep over S; *total_dev points at the device int64 holding R afterwards, *stats_dev at
// 16 device doubles: n, mean[7], M2[7] (sum of squared deviations) of [x, x^2, x^3, max(x-1,0), s, x*s, y].
hipError_t nn_rows_count(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const int64_t** total_dev,
                         const double** stats_dev)
{
    constexpr bool with_stats = true;
    const RowsArgs a = make_args(p, D);
    const size_t n = (size_t)(p.N - 1) * a.ntiles, nwg = rows_nwg(p.M, p.N, a.tchunk);
    int64_t *offs, *rowtot, *rowbase; int32_t* cnt; double *part, *out;
    carve(scratch, n, nwg, p.N, &offs, &rowtot, &rowbase, &cnt, &part, &out);
    const int nrow = p.N - 1 > 0 ? p.N - 1 : 0;
    const dim3 grid(a.ntiles, (p.N - 1 + a.tchunk - 1) / a.tchunk);
    if (n > 0) {
        hipLaunchKernelGGL(rows_count_stats_kernel, grid, dim3(kBlock), 0, st, a, cnt, part);
        // two-level scan: every time step's tiles in their own workgroup, then the per-step totals
        hipLaunchKernelGGL(rows_scan_kernel<int32_t>, dim3((unsigned)nrow), dim3(1024), 0, st, (const int32_t*)cnt, (int64_t)n,
                           (int64_t)a.ntiles, offs, rowtot);
    }
    hipLaunchKernelGGL(rows_scan_kernel<int64_t>, dim3(1), dim3(1024), 0, st, (const int64_t*)rowtot, (int64_t)nrow,
                       (int64_t)(nrow > 0 ? nrow : 1), rowbase, (int64_t*)nullptr);
    if (with_stats) {
        const int np = n > 0 ? (int)(grid.x * grid.y) : 0;
        if (np > kBlock) {  // slices of 256 partials, then the slice results (behind `out` in the scratch)
            const int slices = (np + kBlock - 1) / kBlock;
            double* part2 = out + kTrip;
            hipLaunchKernelGGL(rows_merge_kernel, dim3(slices), dim3(kBlock), 0, st, (const double*)part, np, kBlock, part2);
            hipLaunchKernelGGL(rows_merge_kernel, dim3(1), dim3(kBlock), 0, st, (const double*)part2, slices, slices, out);
        } else {
            hipLaunchKernelGGL(rows_merge_kernel, dim3(1), dim3(kBlock), 0, st, (const double*)part, np, np > 0 ? np : 1, out);
        }
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
        nm.rs[i] = 1.0 / feat_std[i];
        nm.nb[i] = -feat_mean[i] * nm.rs[i];
    }
    nm.rys = 1.0 / y_std;
    nm.nby = -y_mean * nm.rys;
    hipLaunchKernelGGL(rows_write_kernel, dim3(a.ntiles, (p.N - 1 + a.tchunk - 1) / a.tchunk), dim3(kBlock), 0, st, a,
                       nm, offs, rowbase, data, cap);
    return hipGetLastError();
}

}  // namespace omc
