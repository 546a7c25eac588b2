// omc_lsm_persist.hip -- the per-step reference sweep (Options_model.py:108-157, options_model_2.py:278-313,
// polynomial regressor) as ONE persistent launch: every path's state lives on chip for the whole backward
// induction, and the grid-wide dependency of each time step (step t's regression set depends on every
// path's decision at t+1) is an in-launch all-gather of the workgroups' 8 partial moments instead of a
// kernel boundary.
//
// Why: measured with the time-stamping build of lsm_step_kernel, one launch per
// step costs, at 1M paths, 1.3 us of launch gap + 1.6-2.3 us before the first byte of ANY load arrives in
// the new kernel + the prologue's read of 16 KB of partials by every workgroup, and re-reads 13 bytes per
// path and step (S_t again, S_N, the flag) although only S_{t-1} is new.  Here a step reads 4 bytes per
// path (S_{t-1}, prefetched a whole step ahead), keeps S_t in registers, S_N in LDS and the sticky flags in
// a register bit mask, and cash-flows are summed at the moment a path exercises.
//
// Grid: G <= 256 workgroups of 1024 threads, ONE per CU (the launch is refused when the device has fewer
// CUs); thread = NCH chunks of 4 paths (NCH <= 8: up to 8.4M paths per GPU -- BASELINE config 3's shard).
//
// Exchange (cdna_hip_programming.md Guideline 16, form R2 "the data is the flag"): a workgroup publishes its
// 8 doubles as 16 naturally aligned 8-byte granules {tag = epoch, 32 value bits}, each written by ONE
// write-through (sc1, agent-scope relaxed atomic) store; four waves of every workgroup sweep all G x 16
// granules with sc1 loads until every tag equals the epoch -- no fence, no separate flag, no atomics RMW.
// Two granule buffers alternate by epoch parity (a workgroup can be at most one epoch ahead of a reader);
// the buffers are zeroed before every launch and epochs start at 1.  Every spin is bounded by the 100 MHz
// reference clock: on a timeout (workgroups not co-resident, e.g. another process owns CUs) the workgroup
// raises a global error word that every other spin polls, all workgroups leave, and the host falls back to
// the launch-per-step sweep -- the kernel cannot hang.
//
// OUTCOME (MI355X, profiles/persist_stamps_r02g.txt): OFF by default.  The exchange of one epoch -- partials
// published, workgroup 0 sweeps them, solves, publishes the fit, everyone picks it up -- takes 6.4 us at the
// median workgroup (two write-through-store -> remote sc1-load hops at ~3 us each under 255 polling waves)
// against 1.3 us of launch gap + ~2.3 us of cold-start latency for the launch-per-step kernel: 8.5 vs
// 6.2 us per step at 1M paths, 24 vs 21 us at 8M.  An all-gather (every workgroup sweeps all partials) was
// worse still (18 us per step: 1024 polling waves starve the writers).  Kept as an opt-in
// (omc_set_option "step_persistent") and under test; the launch-per-step sweep stays the product path.
//
// Results: the same regression sets, fits and decisions as the launch-per-step sweep (the partial sums of
// a workgroup cover the same paths and are combined over workgroups in the same order); only the order of
// additions INSIDE a wave differs (xor-shuffle tree here, LDS transpose there), i.e. moment sums agree to
// the last bits, not bit for bit.
#include "omc_lsm_dev.h"

namespace omc {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct PersistArgs {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, invK;
    float* sx;
    int32_t* tex;
    float* live;  // state the valuation kernel reads: S_N for a path that never exercised, negative otherwise
    const double* D;
    double* gmom;
    double* betas;
    double* part;             // [8][pstride]: final sums of every workgroup (rows 0..3) for lsm_finalize
    unsigned long long* gran; // [2][nblk][16] granules: every workgroup's partial moments
    unsigned long long* bgran; // [2][16] granules: the fit of the epoch, published by workgroup 0 (8 used per parity)
    unsigned int* err;        // raised by a spin that gave up
    int nblk, pstride, nchunk, write_state;
    unsigned long long spin_ticks;
    unsigned long long* dbg;  // measurement aid (omc_set_option "step_stamps"): [N+1][nblk][8] time stamps
};

constexpr int kPersistBlock = 1024;
constexpr int kPersistWaves = kPersistBlock / 64;
constexpr int kPollWaves = kStepMaxBlocks / 64;  // waves 0..3 sweep the granules of 64 workgroups each
constexpr int kPublishWave = kPersistWaves - 1;  // the last wave publishes

// all 8 accumulators over the wave by an xor-shuffle tree; every lane ends with every total
__device__ __forceinline__ void wave_allreduce8(double (&acc)[8])
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += __shfl_xor(acc[q], off);
    }
}

template <int NCH>
__global__ __launch_bounds__(kPersistBlock) void lsm_sweep_persist_kernel(PersistArgs a)
{
    constexpr bool kPatch = NCH <= 2;  // room for the per-wave LDS transpose patches next to S_N?
    __shared__ float sh_sn[NCH * kPersistBlock * 4];  // terminal spots of this workgroup's paths
    __shared__ double sh_all[kStepMaxBlocks * 8];     // gathered partial moments of one epoch
    __shared__ double wl[kPatch ? kPersistWaves * kWaveRedDoubles : 8];
    __shared__ double sh_w[kPersistWaves * 8];
    __shared__ double sh_beta[4];
    __shared__ int sh_fail;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, G = a.nblk, is_put = a.is_put;
    const double K = a.K, invK = a.invK;
    const int64_t cstride = (int64_t)G * kPersistBlock * 4;
    const int64_t j0 = ((int64_t)blockIdx.x * kPersistBlock + tid) * 4;
    gu64* const gran = (gu64*)a.gran;
    gu64* const bgran = (gu64*)a.bgran;
    gu32* const err = (gu32*)a.err;
    const bool leader = blockIdx.x == 0;
    if (tid == 0) sh_fail = 0;
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mark = [&](int k) {
        if (a.dbg) {
            __builtin_amdgcn_sched_barrier(0);
            stamp[k] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float st[NCH][4], sm[NCH][4];
    bool valid[NCH];
    uint32_t exbits = 0;  // bit 4c+v: path v of chunk c has exercised
    double fin[4] = {0.0, 0.0, 0.0, 0.0};  // sum, sumsq, n_exercised, n_zero of the t = dt cash-flows

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        valid[c] = c < a.nchunk && j0 + c * cstride < a.M;
#pragma unroll
        for (int v = 0; v < 4; ++v) st[c][v] = sm[c][v] = 0.f;
    }
    auto load_row = [&](float (&dst)[NCH][4], int t) {
        const float* row = a.S + (int64_t)t * a.ld + j0;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (valid[c]) loadf<4>(row + c * cstride, dst[c]);
    };
    load_row(st, N);
    if (N >= 2) load_row(sm, N - 1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int v = 0; v < 4; ++v) sh_sn[(c * kPersistBlock + tid) * 4 + v] = st[c][v];  // own slots only
    }

    // moments of step tm from the row in `sm`, the sticky flags and the terminal payoffs -> published
    // under `epoch`
    auto moments_and_publish = [&](int tm, unsigned epoch) {
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.0;
        bool added = false;
        const double Dm = a.D[N - tm];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (!valid[c]) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const double imm = payoff_d(sm[c][v], K, is_put);
                if (imm > 0.0 && ((exbits >> (4 * c + v)) & 1u) == 0u) {
                    double p = payoff_d(sh_sn[(c * kPersistBlock + tid) * 4 + v], K, is_put);
                    p = p > 0.0 ? p : 0.0;
                    accumulate_moments(acc, fma((double)sm[c][v], invK, -1.0), p * Dm);
                    added = true;
                }
            }
        }
        mark(6);
        if constexpr (kPatch) {
            double s = 0.0;
            if (__builtin_amdgcn_ballot_w64(added) != 0) s = wave_reduce8(acc, wl + wave * kWaveRedDoubles);
            if ((lane & 7) == 0) sh_w[wave * 8 + (lane >> 3)] = s;
        } else {
            if (__builtin_amdgcn_ballot_w64(added) != 0) {
                wave_allreduce8(acc);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = 0.0;
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) sh_w[wave * 8 + q] = acc[q];
            }
        }
        __syncthreads();
        // Published by a wave that does NOT poll: on gfx9 a wave's loads and stores share one in-order
        // counter, so a poller that had just stored would sit behind its own write-through stores
        // (measured: ~5 us each) on its next counted wait.
        if (wave == kPublishWave && lane < 8) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kPersistWaves; ++w) tot += sh_w[w * 8 + lane];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(tot);
            gu64* g = gran + ((size_t)(epoch & 1u) * G + blockIdx.x) * 16 + 2 * lane;
            const unsigned long long tag = (unsigned long long)epoch << 32;
            __hip_atomic_store(g, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // One bounded poll loop for both roles.  `ready(x)` says whether this lane's granules carry `epoch`.
    // Returns false when a spin gave up somewhere (this one by time, or another one through `err`).
    auto spin_until = [&](auto&& load_and_check, bool participate) -> bool {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned polls = 1;; ++polls) {
            const bool ok = load_and_check();
            if (__builtin_amdgcn_ballot_w64(participate && !ok) == 0) {
                if (wave == 0) stamp[2] = polls;
                return true;
            }
            const unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool late = __builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks;
            if (e != 0u || late) {
                if (late && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    };

    // The exchange of one epoch, two hops: every workgroup has published its 8 partial moments; workgroup 0
    // sweeps them all (256 lines per pass, ONE reader: no poller storm on the lines the writers are trying to
    // reach), combines them in a fixed order, solves, and publishes the fit (b0, b1, b2, n) as 8 granules in
    // one line; every other workgroup polls that one line with one wave.  Leaves the fit in sh_beta; false:
    // leave the kernel.
    auto exchange = [&](unsigned epoch, int t) -> bool {
        if (leader) {
            if (wave < kPollWaves) {
                const int b = wave * 64 + lane;
                const bool mine = b < G;
                const gu64* g = gran + ((size_t)(epoch & 1u) * G + (mine ? b : 0)) * 16;
                unsigned long long x[16];
                const bool fine = spin_until([&]() {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        x[k] = __hip_atomic_load(g + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (unsigned)(x[k] >> 32) == epoch;
                    }
                    return ok;
                }, mine);
                if (!fine) {
                    sh_fail = 1;
                } else if (mine) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned long long bits = (x[2 * q] << 32) | (x[2 * q + 1] & 0xffffffffull);
                        sh_all[b * 8 + q] = __longlong_as_double((long long)bits);
                    }
                }
            }
            __syncthreads();
            if (sh_fail) return false;
            if (wave >= 8) {  // wave 8 + q totals quantity q over the workgroups
                const int q = wave - 8;
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < kStepMaxBlocks / 64; ++i) {
                    const int b = lane + 64 * i;
                    s += b < G ? sh_all[b * 8 + q] : 0.0;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
                if (lane == 0) sh_w[q] = s;
            }
            __syncthreads();
            if (wave == kPublishWave) {
                double m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = sh_w[q];
                double beta[3];
                solve_poly2(m, beta);
                const double f4[4] = {beta[0], beta[1], beta[2], m[0]};
                if (lane < 8) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(f4[lane >> 1]);
                    const unsigned long long word = (lane & 1) ? (bits & 0xffffffffull) : (bits >> 32);
                    __hip_atomic_store(bgran + (size_t)(epoch & 1u) * 16 + lane, ((unsigned long long)epoch << 32) | word,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane == 0) {
                    sh_beta[0] = f4[0]; sh_beta[1] = f4[1]; sh_beta[2] = f4[2]; sh_beta[3] = f4[3];
                    double* bo = a.betas + (size_t)t * 4;
                    bo[0] = f4[0]; bo[1] = f4[1]; bo[2] = f4[2]; bo[3] = f4[3];
#pragma unroll
                    for (int q = 0; q < 8; ++q) a.gmom[(size_t)t * 8 + q] = m[q];
                }
            }
            __syncthreads();
            return true;
        }
        if (wave == 0) {
            const bool mine = lane < 8;
            unsigned long long x = 0;
            const gu64* g = bgran + (size_t)(epoch & 1u) * 16 + (mine ? lane : 0);
            const bool fine = spin_until([&]() {
                x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return (unsigned)(x >> 32) == epoch;
            }, mine);
            if (!fine) {
                sh_fail = 1;
            } else {
                const unsigned lo = (unsigned)__shfl(x, lane | 1), hi = (unsigned)__shfl(x, lane & ~1);
                if (mine && (lane & 1) == 0)
                    sh_beta[lane >> 1] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
        }
        __syncthreads();
        return sh_fail == 0;
    };

    unsigned epoch = 0;
    if (N >= 2) moments_and_publish(N - 1, ++epoch);  // what launch t = N of the per-step sweep does

    for (int t = N - 1; t >= 1; --t) {
        // S_t moves into place
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) st[c][v] = sm[c][v];
        // S_{t-1} starts its journey now and is needed only after the exchange and the apply phase; the
        // polling waves start theirs after the gather (their granule loads would otherwise wait behind it)
        const bool poller = leader ? wave < kPollWaves : wave == 0;
        if (t >= 2 && !poller) load_row(sm, t - 1);
        mark(0);
        if (!exchange(epoch, t)) return;  // uniform: every thread saw the same sh_fail after a barrier
        if (t >= 2 && poller) load_row(sm, t - 1);
        mark(1);
        mark(3);
        mark(4);
        const double b0 = sh_beta[0], b1 = sh_beta[1], b2 = sh_beta[2];
        if (sh_beta[3] > 0.5) {
            const double Dt = a.D[t - 1];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (!valid[c]) continue;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double imm = payoff_d(st[c][v], K, is_put);
                    if (imm > 0.0 && ((exbits >> (4 * c + v)) & 1u) == 0u) {
                        const double u = fma((double)st[c][v], invK, -1.0);
                        const double cont = fma(u, fma(u, b2, b1), b0);
                        if (imm > cont) {  // at most once per path
                            exbits |= 1u << (4 * c + v);
                            const double cf = imm * Dt;
                            fin[0] += cf;
                            fin[1] += cf * cf;
                            fin[2] += 1.0;
                            if (a.write_state) {
                                a.sx[j0 + c * cstride + v] = st[c][v];
                                a.tex[j0 + c * cstride + v] = t;
                            }
                        }
                    }
                }
            }
        }
        mark(5);
        if (t >= 2) moments_and_publish(t - 1, ++epoch);
        mark(7);
        if (a.dbg && tid == 0) {
            unsigned long long* d = a.dbg + ((size_t)(N - t) * G + blockIdx.x) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = stamp[k];
        }
    }

    // paths that never exercised keep their terminal payoff, valued at t = dt
    {
        const double Dn = a.D[N - 1];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (!valid[c]) continue;
            uint32_t flags = 0;
            float sn4[4];
            int32_t tn4[4];
            bool any = false;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool ex = (exbits >> (4 * c + v)) & 1u;
                sn4[v] = sh_sn[(c * kPersistBlock + tid) * 4 + v];
                tn4[v] = N;
                if (ex) {
                    flags |= 1u << (8 * v);
                } else {
                    double p = payoff_d(sn4[v], K, is_put);
                    p = p > 0.0 ? p : 0.0;
                    const double cf = p * Dn;
                    fin[0] += cf;
                    fin[1] += cf * cf;
                    fin[3] += (cf == 0.0) ? 1.0 : 0.0;
                    any = true;
                }
            }
            if (a.write_state) {
                {
                    float lv[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) lv[v] = ((flags >> (8 * v)) & 0xffu) ? -1.0f : sn4[v];
                    *reinterpret_cast<float4*>(a.live + j0 + c * cstride) = make_float4(lv[0], lv[1], lv[2], lv[3]);
                }
                if (any) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        if (((flags >> (8 * v)) & 0xffu) == 0u) {
                            a.sx[j0 + c * cstride + v] = sn4[v];
                            a.tex[j0 + c * cstride + v] = tn4[v];
                        }
                    }
                }
            }
        }
        double acc[8] = {fin[0], fin[1], fin[2], fin[3], 0.0, 0.0, 0.0, 0.0};
        wave_allreduce8(acc);
        __syncthreads();  // sh_w may still be read by the last publish
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sh_w[wave * 8 + q] = acc[q];
        }
        __syncthreads();
        if (tid < 4) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kPersistWaves; ++w) tot += sh_w[w * 8 + tid];
            a.part[(size_t)tid * a.pstride + blockIdx.x] = tot;
        }
    }
}

// result[q] for the four sums; result[7] = error word (non-zero: the sweep gave up, results invalid)
__global__ __launch_bounds__(kBlock) void lsm_persist_finalize_kernel(const double* part, const double* gmom,
                                                                      const unsigned int* err, double* result,
                                                                      int nblk, int N, int pstride)
{
    lsm_finalize_body(part, gmom, result, nblk, N, pstride);
    __syncthreads();  // result[7] was just written (as 0) by another thread of this workgroup
    if (threadIdx.x == 0) result[7] = (double)*err;
}

size_t lsm_persist_scratch_bytes() { return sizeof(unsigned long long) * (2 * kStepMaxBlocks * 16 + 2 * 16) + 256; }

int lsm_persist_max_chunks() { return 8; }

bool lsm_persist_supported(const LsmProblem& p, int device_cus)
{
    if (p.N < 1 || (p.M % 4) != 0 || (p.ld % 4) != 0 || ((uintptr_t)p.S % 16) != 0) return false;
    const int G = lsm_sweep_blocks(p.M);
    if (lsm_step_block_threads() != kPersistBlock || G > device_cus) return false;
    const int64_t per_sweep = (int64_t)G * kPersistBlock * 4;
    return (p.M + per_sweep - 1) / per_sweep <= lsm_persist_max_chunks();
}

// scratch: lsm_persist_scratch_bytes() of device memory (granules, then the error word)
hipError_t lsm_sweep_persistent(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch,
                                bool write_state, double spin_seconds)
{
    PersistArgs a;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put; a.K = p.K; a.invK = 1.0 / p.K;
    a.sx = w.sx; a.tex = w.tex; a.live = w.live; a.D = w.D; a.gmom = w.gmom; a.betas = w.betas; a.part = w.part;
    a.gran = (unsigned long long*)scratch;
    a.bgran = a.gran + 2 * kStepMaxBlocks * 16;
    a.err = (unsigned int*)(a.bgran + 2 * 16);
    a.nblk = lsm_sweep_blocks(p.M);
    a.pstride = kMaxLsmBlocks;
    const int64_t per_sweep = (int64_t)a.nblk * kPersistBlock * 4;
    a.nchunk = (int)((p.M + per_sweep - 1) / per_sweep);
    a.write_state = write_state ? 1 : 0;
    a.spin_ticks = (unsigned long long)(spin_seconds * 1e8);
    a.dbg = w.dbg;
    hipError_t e = hipMemsetAsync(scratch, 0, lsm_persist_scratch_bytes(), st);  // tags and error word: every launch
    if (e != hipSuccess) return e;
    const dim3 grid(a.nblk), block(kPersistBlock);
    if (a.nchunk <= 1) hipLaunchKernelGGL((lsm_sweep_persist_kernel<1>), grid, block, 0, st, a);
    else if (a.nchunk <= 2) hipLaunchKernelGGL((lsm_sweep_persist_kernel<2>), grid, block, 0, st, a);
    else if (a.nchunk <= 4) hipLaunchKernelGGL((lsm_sweep_persist_kernel<4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((lsm_sweep_persist_kernel<8>), grid, block, 0, st, a);
    hipLaunchKernelGGL(lsm_persist_finalize_kernel, dim3(1), dim3(kBlock), 0, st, w.part, w.gmom, a.err, w.result,
                       a.nblk, p.N, kMaxLsmBlocks);
    return hipGetLastError();
}

}  // namespace omc
