// omc_contnet_dev.h -- device code of the per-step ContNet regressor of the reference's v1 / v2 pricers
// (Options_model.py:112-151, options_model_2.py:283-312; see omc_contnet.hip for the flow).  Bodies are __device__
// functions shared by the single-pricing launchers (omc_contnet.hip) and the batched many-pricings launchers
// (omc_batch.hip: problem index on the grid); a body only looks at blockIdx.x.
#pragma once
#include "omc_device.h"
#include "omc_kernels.h"

namespace omc {

constexpr int kCnBlock = 256;
constexpr int kCnChunks = 8;                       // chunks of 256 consecutive paths per workgroup
constexpr int kCnSpan = kCnBlock * kCnChunks;      // paths per workgroup

struct CnArgs {
    const float* St;    // row t of the path matrix
    const float* SN;    // row N
    const float* live;  // sticky state of the per-step sweep: negative = has exercised
    int64_t M;
    double K, Dt;       // Dt = exp(-r dt (N - t)): terminal payoff valued at t
    int is_put, nblk;
    int32_t* cnt;       // [nblk]
    double* s1;         // [nblk] sum (S - K)
    double* s2;         // [nblk] sum (S - K)^2
    int64_t* offs;      // [nblk + 1]
    double* hdr;        // n, mean, 1/std (1 when std == 0), std
    float* data;        // [n][8]
    float* cont;        // [M] continuation values of the members (others untouched)
};

__device__ __forceinline__ bool member(const CnArgs& a, int64_t j, float s)
{
    return payoff_d(s, a.K, a.is_put) > 0.0 && !(a.live[j] < 0.0f);
}

// fixed-order sum of one double per thread over the workgroup (valid in thread 0)
__device__ __forceinline__ double block_sum(double v, double* sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (tid == 0)
        for (int w = 0; w < kCnBlock / 64; ++w) s += sh[w];
    return s;
}

__device__ __forceinline__ void cn_count_body(const CnArgs& a)
{
    __shared__ double sh[kCnBlock / 64];
    __shared__ int shc[kCnBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kCnSpan;
    int c = 0;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < kCnChunks; ++k) {
        const int64_t j = base + (int64_t)k * kCnBlock + tid;
        if (j < a.M) {
            const float s = a.St[j];
            if (member(a, j, s)) {
                const double d = (double)s - a.K;
                ++c;
                s1 += d;
                s2 += d * d;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane == 0) shc[wave] = c;
    const double t1 = block_sum(s1, sh);
    const double t2 = block_sum(s2, sh);
    if (tid == 0) {
        int ct = 0;
        for (int w = 0; w < kCnBlock / 64; ++w) ct += shc[w];
        a.cnt[blockIdx.x] = ct;
        a.s1[blockIdx.x] = t1;
        a.s2[blockIdx.x] = t2;
    }
}

// one workgroup of 1024 threads: exclusive scan of cnt, totals -> hdr
__device__ __forceinline__ void cn_scan_body(const CnArgs& a)
{
    __shared__ int64_t seg[1024];
    __shared__ double r1[1024], r2[1024];
    const int tid = threadIdx.x;
    const int n = a.nblk;
    const int len = (n + 1023) / 1024, lo = tid * len, hi = lo + len < n ? lo + len : n;
    int64_t s = 0;
    double q1 = 0.0, q2 = 0.0;
    for (int i = lo; i < hi; ++i) {
        s += a.cnt[i];
        q1 += a.s1[i];
        q2 += a.s2[i];
    }
    seg[tid] = s;
    r1[tid] = q1;
    r2[tid] = q2;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int64_t v = tid >= d ? seg[tid - d] : 0;
        __syncthreads();
        seg[tid] += v;
        __syncthreads();
    }
    for (int d = 512; d >= 1; d >>= 1) {  // fixed tree
        if (tid < d) {
            r1[tid] += r1[tid + d];
            r2[tid] += r2[tid + d];
        }
        __syncthreads();
    }
    int64_t run = tid ? seg[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        a.offs[i] = run;
        run += a.cnt[i];
    }
    if (tid == 0) {
        const int64_t R = seg[1023];
        a.offs[n] = R;
        double mean = a.K, sd = 0.0;
        if (R > 0) {
            const double m1 = r1[0] / (double)R;
            double var = r2[0] / (double)R - m1 * m1;  // population variance (numpy's X.std())
            if (!(var > 0.0)) var = 0.0;
            mean = a.K + m1;
            sd = sqrt(var);
        }
        a.hdr[0] = (double)R;
        a.hdr[1] = mean;
        a.hdr[2] = sd > 0.0 ? 1.0 / sd : 1.0;  // X.std() > 0 else X - X.mean()
        a.hdr[3] = sd;
    }
}

// rank of this thread's member among the members of the workgroup's span, in path order
__device__ __forceinline__ void cn_rows_body(const CnArgs& a)
{
    __shared__ int wcnt[kCnBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kCnSpan;
    const double mean = a.hdr[1], rs = a.hdr[2];
    int64_t run = a.offs[blockIdx.x];
    for (int k = 0; k < kCnChunks; ++k) {
        const int64_t j = base + (int64_t)k * kCnBlock + tid;
        float s = 0.0f;
        bool f = false;
        if (j < a.M) {
            s = a.St[j];
            f = member(a, j, s);
        }
        const uint64_t bal = __builtin_amdgcn_ballot_w64(f);
        if (lane == 0) wcnt[wave] = __builtin_popcountll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kCnBlock / 64; ++w) {
            before += w < wave ? wcnt[w] : 0;
            total += wcnt[w];
        }
        if (f) {
            const int64_t row = run + before + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
            double y = payoff_d(a.SN[j], a.K, a.is_put);
            y = y > 0.0 ? y * a.Dt : 0.0;
            float4* out = reinterpret_cast<float4*>(a.data + row * 8);
            out[0] = make_float4((float)(((double)s - mean) * rs), 0.0f, 0.0f, 0.0f);
            out[1] = make_float4(0.0f, 0.0f, 0.0f, (float)y);
        }
        run += total;
        __syncthreads();
    }
}

struct CnInitArgs {
    float* params;
    float* m;
    float* v;
    int H, h, np;      // padded width (trainer granularity), actual width, parameter count of the padded net
    uint32_t k0, k1, t;
};

__device__ __forceinline__ float uniform_pm(uint32_t bits, float bound)
{
    // 24 random bits -> [-bound, bound)
    return ((float)(bits >> 8) * (1.0f / 8388608.0f) - 1.0f) * bound;
}

// parameter i of step t's fresh net (and zeroed Adam moments); `i` = flat index in the padded layout
__device__ __forceinline__ void cn_init_one(const CnInitArgs& a, const int i)
{
    if (i >= a.np) return;
    const int H = a.H, h = a.h, conn = H * H + H;
    const float bh = 1.0f / sqrtf((float)h);
    float bound = 0.0f;
    if (i < H * 8) {  // layer 0: [unit][8] = weight of the one input, 6 unused inputs, bias
        const int unit = i >> 3, col = i & 7;
        if (unit < h && (col == 0 || col == 7)) bound = 1.0f;  // fan_in = 1
    } else if (i < H * 8 + H * H) {
        const int e = i - H * 8, out = e / H, in = e % H;
        if (out < h && in < h) bound = bh;
    } else if (i < H * 8 + conn) {
        if (i - (H * 8 + H * H) < h) bound = bh;
    } else if (i < H * 8 + conn + H) {
        if (i - (H * 8 + conn) < h) bound = bh;
    } else {
        bound = bh;  // output bias
    }
    float val = 0.0f;
    if (bound > 0.0f) {
        const U4 o = philox4x32_10((uint32_t)i, a.t, 0x434e4554u /* "CNET" */, 0u, a.k0, a.k1);
        val = uniform_pm(o.x, bound);
    }
    a.params[i] = val;
    a.m[i] = 0.0f;
    a.v[i] = 0.0f;
}

struct CnFwdArgs {
    CnArgs c;
    const float* params;
    int h;
};

// continuation value of every member: the network in float32 like the reference's net(X_tensor)
template <int H>
__device__ __forceinline__ void cn_forward_body(const CnFwdArgs& fa)
{
    constexpr int NP = H * 8 + H * H + H + H + 1;
    extern __shared__ float sp[];
    const CnArgs& a = fa.c;
    stage_block(fa.params, NP, (int)threadIdx.x, [&](int i, float v) { sp[i] = v; });  // kCnBlock == 256
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * kCnBlock + threadIdx.x;
    if (j >= a.M) return;
    const float s = a.St[j];
    if (!member(a, j, s)) return;
    const float xs = (float)(((double)s - a.hdr[1]) * a.hdr[2]);
    const int h = fa.h;
    float h1[H];
#pragma unroll
    for (int k = 0; k < H; ++k) h1[k] = fmaxf(fmaf(sp[k * 8], xs, sp[k * 8 + 7]), 0.0f);
    const float* W1 = sp + H * 8;
    const float* b1 = W1 + H * H;
    const float* wo = b1 + H;
    float o = wo[H];
    for (int i = 0; i < h; ++i) {  // units >= h are identically zero
        float z = b1[i];
        const float4* wr = reinterpret_cast<const float4*>(W1 + i * H);
#pragma unroll
        for (int k = 0; k < H / 4; ++k) {
            const float4 w4 = wr[k];
            z = fmaf(w4.x, h1[4 * k], z);
            z = fmaf(w4.y, h1[4 * k + 1], z);
            z = fmaf(w4.z, h1[4 * k + 2], z);
            z = fmaf(w4.w, h1[4 * k + 3], z);
        }
        o = fmaf(wo[i], fmaxf(z, 0.0f), o);
    }
    a.cont[j] = o;
}

}  // namespace omc
