// omc_device.h -- device-side building blocks shared by the gfx950 kernels:
// Philox4x32-10 counter RNG, Box-Muller on the hardware transcendental units,
// block-wide reduction of 8 double accumulators through LDS, 3x3 normal-equation solve.
//
// Wave = 64 lanes (gfx950).  Everything here is written for 256-thread workgroups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace omc {

constexpr int kBlock = 256;        // threads per workgroup (4 waves, one per SIMD)
constexpr int kRedStride = 264;    // doubles per quantity row in LDS: 256 + 8 pad ->
                                   // ds_read_b64 by (q,sub) lanes lands on distinct banks
constexpr int kNQ = 8;             // quantities reduced together

// ------------------------------------------------------------------ Philox4x32-10
// Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3" (SC'11).
// Round keys are wave-uniform, so they stay in SGPRs; each round is two 32x32->64
// multiplies (v_mad_u64_u32 / v_mul_hi_u32) plus three XORs per lane.
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// ------------------------------------------------------------------ Box-Muller
// u1 in (0,1], u2 in [0,1) built from the top 24 bits of each word (exact in f32).
// v_log_f32 is log2, v_sin_f32 / v_cos_f32 take their argument in revolutions, so the
// 2*pi never has to be multiplied in: radius = sqrt(-2 ln2 * log2(u1)), angle = u2 turns.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& zc, float& zs)
{
    const float u1 = __builtin_fmaf((float)(a >> 8), 0x1p-24f, 0x1p-25f);
    const float u2 = (float)(b >> 8) * 0x1p-24f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    zc = rad * __builtin_amdgcn_cosf(u2);
    zs = rad * __builtin_amdgcn_sinf(u2);
}

// four N(0,1) draws of one Philox block; counter = (pair_lo, pair_hi, block, stream)
__device__ __forceinline__ void normals4(uint64_t pair, uint32_t block, uint32_t stream,
                                         uint32_t k0, uint32_t k1, float (&z)[4])
{
    const U4 o = philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), block, stream, k0, k1);
    box_muller(o.x, o.y, z[0], z[1]);
    box_muller(o.z, o.w, z[2], z[3]);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------ XCD-aware block numbering
// A launch's workgroups go to the chip's 8 XCDs round robin in launch order (blockIdx.x % 8 for a 1-D grid).
// xcd_block() renumbers the G blocks of a grid row so that the blocks running on ONE XCD own CONSECUTIVE numbers:
// a bijection on [0, G) (residue r gets the r-th run, of G/8 or G/8 + 1 numbers).
__device__ __forceinline__ int xcd_block(int bx, int G)
{
    const int r = bx & 7, q = G >> 3, rem = G & 7;
    return r * q + (r < rem ? r : rem) + (bx >> 3);
}

// ------------------------------------------------------------------ staging a parameter block into LDS
// put(i, src[i]) for i in [0, n), 256-thread workgroups.  A plain `for (i = tid; i < n; i += 256) dst[f(i)] = src[i]`
// compiles to load -> s_waitcnt vmcnt(0) -> ds_write per (pair of) iteration(s): one global round trip after the
// other -- 8 of them for a 64 x 64 connection, ~10 us at the head of every launch of the trainer (round 4, seen in
// the ISA).  Here a thread's loads of a chunk (up to 16) are all issued before the first store, so a chunk costs one
// round trip.  n is a compile-time constant at every call site: the bounds checks fold away.
template <class Put>
__device__ __forceinline__ void stage_block(const float* __restrict__ src, const int n, const int tid, Put&& put)
{
    __builtin_assume(tid >= 0 && tid < 256);
    for (int base = 0; base < n; base += 256 * 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = base + tid + 256 * k;
            v[k] = i < n ? src[i] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = base + tid + 256 * k;
            if (i < n) put(i, v[k]);
        }
    }
}


// ------------------------------------------------------------------ sums inside groups of 8 lanes
// s + the value of lane ^ 1, ^ 2, ^ 4 in turn, as data-parallel-primitive moves in the vector pipe (quad permutes, then the
// mirror of a half row: after the first two steps the four lanes of a quad agree, so lane 7 - i holds what lane i ^ 4
// holds) -- the same operand pairs, hence the same bits, as three __shfl_xor, without their three LDS round trips.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_group8(double s)
{
    s += dpp_move<0xB1>(s);   // quad_perm [1,0,3,2]
    s += dpp_move<0x4E>(s);   // quad_perm [2,3,0,1]
    s += dpp_move<0x141>(s);  // row_half_mirror
    return s;
}

// ------------------------------------------------------------------ block reduction
// Sums acc[0..7] over the 256 threads of the block in a fixed order (bitwise
// reproducible run to run).  `red` is LDS, kNQ*kRedStride doubles.  On return lanes of
// wave 0 with (lane & 7) == 0 hold the total of quantity lane >> 3 in the return value;
// other threads get garbage.  The caller decides where the totals go.  One barrier
// inside; the caller must separate two uses of the same `red` by another barrier.
__device__ __forceinline__ double block_reduce8(const double (&acc)[kNQ], double* red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) red[q * kRedStride + tid] = acc[q];
    __syncthreads();
    double s = 0.0;
    if (tid < 64) {
        const int q = tid >> 3, sub = tid & 7;
        const double* p = red + q * kRedStride + sub;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) s += p[8 * i];
        s = sum_group8(s);
    }
    return s;
}

// ------------------------------------------------------------------ wave reduction
// Same job for ONE wave, no workgroup barrier: the 64 lanes transpose their 8 accumulators
// through a wave-private LDS patch (DS instructions of a wave execute in order), each lane
// then owns 1/8 of one quantity, and three xor-shuffles finish inside groups of 8 lanes.
// ~8 ds_write_b64 + 8 ds_read_b64 + 3 in-register exchanges (sum_group8) instead of 48 shuffles.  Row stride 72 doubles
// puts the (q, sub) read pattern on distinct banks.  Every lane returns the total of quantity
// lane >> 3.
constexpr int kWaveRedStride = 72;
constexpr int kWaveRedDoubles = kNQ * kWaveRedStride;

__device__ __forceinline__ double wave_reduce8(const double (&acc)[kNQ], double* wl)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) wl[q * kWaveRedStride + lane] = acc[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* p = wl + (lane >> 3) * kWaveRedStride + (lane & 7);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[8 * i];
    s = sum_group8(s);
    __builtin_amdgcn_wave_barrier();
    return s;
}

// ------------------------------------------------------------------ 3x3 OLS solve
// 1/a for a normal, positive a: v_rcp_f64 seed + two Newton steps (~1 ulp).  An IEEE division
// expands to ~15 dependent instructions; the per-step kernel solves on its critical path.
__device__ __forceinline__ double rcp_nr(double a)
{
    double x = __builtin_amdgcn_rcp(a);
    x = fma(fma(-a, x, 1.0), x, x);
    x = fma(fma(-a, x, 1.0), x, x);
    return x;
}

// y ~ b0 + b1 u + b2 u^2 from m = {n, Su, Su2, Su3, Su4, Sy, Suy, Su2y}; centred LDL^T,
// degree reduced to n-1 for n < 3 or when a pivot is not safely positive.  Three reciprocals
// (1/n, 1/c11, 1/d2) instead of eight divisions.
__device__ __forceinline__ void solve_poly2(const double (&m)[8], double (&beta)[3])
{
    const double n = m[0];
    beta[0] = beta[1] = beta[2] = 0.0;
    if (n < 0.5) return;
    const double rn = rcp_nr(n);
    const double mu = m[1] * rn, my = m[5] * rn;
    const double c11 = m[2] - m[1] * mu;
    const double c1y = m[6] - m[1] * my;
    if (n < 1.5 || !(c11 > 1e-14 * fabs(m[2]) + 1e-300)) { beta[0] = my; return; }
    const double mq = m[2] * rn;
    const double c22 = m[4] - m[2] * mq;
    const double c12 = m[3] - m[1] * mq;
    const double c2y = m[7] - m[2] * my;
    const double r11 = rcp_nr(c11);
    const double l21 = c12 * r11;
    const double d2 = c22 - l21 * c12;
    if (n < 2.5 || !(d2 > 1e-12 * fabs(c22) + 1e-300)) {
        beta[1] = c1y * r11;
        beta[0] = my - beta[1] * mu;
        return;
    }
    const double b2 = (c2y - l21 * c1y) * rcp_nr(d2);
    const double b1 = (c1y - c12 * b2) * r11;
    beta[2] = b2;
    beta[1] = b1;
    beta[0] = my - b1 * mu - b2 * mq;
}

__device__ __forceinline__ double payoff_d(float s, double K, int is_put)
{
    return is_put ? K - (double)s : (double)s - K;
}

__device__ __forceinline__ void accumulate_moments(double (&acc)[8], double u, double y)
{
    const double u2 = u * u;
    acc[0] += 1.0;
    acc[1] += u;
    acc[2] += u2;
    acc[3] += u2 * u;
    acc[4] += u2 * u2;
    acc[5] += y;
    acc[6] += u * y;
    acc[7] += u2 * y;
}

}  // namespace omc
