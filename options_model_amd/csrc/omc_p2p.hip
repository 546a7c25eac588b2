// omc_p2p.hip -- the per-step moment exchange WITHOUT a collective library.
//
// The per-step flows need, after every time step, the sum over the ranks of 8 doubles per pricing (8K for K pricings
// per launch).  A ring all-reduce of 64 bytes is pure latency (and a kernel or two of the collective library per
// step); on one node every GPU can instead WRITE its contribution straight into every peer's memory over xGMI and
// each rank sums what has arrived itself:
//   * every rank owns a small mailbox in fine-grained device memory, exported with hipIpcGetMemHandle and mapped by
//     all peers (hipIpcOpenMemHandle): data[parity][rank][256 doubles] + flag[parity][rank][32 pricings];
//   * one kernel per step and rank (one workgroup per pricing): reduce this rank's partial moments (fixed order) ->
//     store the 8 sums into the slot [parity][my rank] of EVERY rank's mailbox (system-scope stores), release, store
//     the exchange's epoch into the slot's flag -> poll the own mailbox's flags of all ranks for this epoch
//     (system-scope loads, bounded by a deadline) -> add the contributions in RANK order (every rank gets the same
//     bits) -> gmom[t];
//   * two parities: a rank can run at most one exchange ahead of a peer (it cannot finish exchange e + 1 without that
//     peer's contribution to e + 1, which the peer sends only after it has read everything of exchange e), so slot
//     e & 1 is never overwritten before it has been read.  Epochs only grow: nothing is ever reset.
// Bounded: a flag that does not arrive within the deadline sets a sticky error word in the rank's own memory (later
// exchanges return at once), the sums become NaN, the host reports an error -- never a hang.
// COLLECTIVE failure: a rank whose error word is set publishes a POISON flag (the largest epoch) instead of its
// epoch into every peer's mailbox, in both parities, from then on; a peer that reads it sets its own error word
// (and poisons in turn).  A rank that is merely slow -- alive, but past a peer's deadline -- can therefore not leave
// the others with finite sums built on a contribution that was given up on: the next exchange fails everywhere.
// What a poison cannot reach any more (the failure happened in a pricing's LAST exchange) is caught by the host:
// every rank adds its error word to slot 6 of the result sums before their all-reduce (p2p_stamp_results).
// The reference has no counterpart (no distributed code at all).
#include "omc_lsm_dev.h"
#include "omc_p2p.h"

#include <cstring>

namespace omc {

namespace {

struct Mailbox {
    double data[2][kP2PMaxWorld][kP2PSlotDoubles];
    unsigned long long flag[2][kP2PMaxWorld][kP2PMaxPricings];
    unsigned long long error;  // sticky: an exchange timed out on this rank
};

struct P2PJob {  // one pricing: where its partials are and where the global moments go
    const double* part;
    double* gmom;
    int nblk, pstride, gstride, pad;
};

constexpr unsigned long long kPoison = ~0ull;  // epochs only grow and never get here

struct P2PArgs {
    Mailbox* box[kP2PMaxWorld];  // every rank's mailbox as mapped in THIS process (own included)
    int rank, world, t, njobs;
    unsigned long long epoch, deadline_ticks;  // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ void exchange_body(const P2PJob& j, const P2PArgs& a, const int k)
{
    __shared__ double red[kNQ * kRedStride];
    __shared__ double loc[8];
    __shared__ int arrived;
    const int tid = threadIdx.x;
    const int par = (int)(a.epoch & 1ull);
    Mailbox* own = a.box[a.rank];
    // 1. this rank's moments of step t for pricing k: the partial slabs in index order (lsm_reduce_step_body's sums)
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    const double* pp = j.part + (size_t)(a.t & 1) * 8 * j.pstride;
    for (int i = tid; i < j.nblk; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += pp[q * j.pstride + i];
    }
    const double s = block_reduce8(acc, red);
    if (tid < 64 && (tid & 7) == 0) loc[tid >> 3] = s;
    __syncthreads();
    // 2. wave 0 publishes: lane = (peer, quantity) -> that peer's slot of THIS rank; then the flags
    if (tid < 64) {
        for (int i = tid; i < a.world * 8; i += 64) {
            const int r = i >> 3, q = i & 7;
            __hip_atomic_store(&a.box[r]->data[par][a.rank][k * 8 + q], loc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the wave's stores above have left before the flags do
        const bool dead = __hip_atomic_load(&own->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
        if (tid < a.world)
            __hip_atomic_store(&a.box[tid]->flag[par][a.rank][k], dead ? kPoison : a.epoch, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        // 3. ... and waits for every rank's flag in its own mailbox (bounded)
        bool ok = true, poisoned = false;
        if (tid < a.world) {
            ok = false;
            if (!dead) {
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    const unsigned long long f =
                        __hip_atomic_load(&own->flag[par][tid][k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (f == kPoison) {
                        poisoned = true;
                        break;
                    }
                    if (f >= a.epoch) {
                        ok = true;
                        break;
                    }
                    if (wall_clock64() - t0 > a.deadline_ticks) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        const bool all = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
        if (!all) {
            // 1 = own deadline ran out, 2 = a peer had given up; sticky
            const bool any_poison = __builtin_amdgcn_ballot_w64(poisoned) != 0ull;
            if (tid == 0 && !dead)
                __hip_atomic_store(&own->error, any_poison ? 2ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < a.world) {  // tell every peer, in the slot of this exchange AND of the next one
                __hip_atomic_store(&a.box[tid]->flag[par][a.rank][k], kPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&a.box[tid]->flag[par ^ 1][a.rank][k], kPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        if (tid == 0) arrived = all ? 1 : 0;
    }
    __syncthreads();
    // 4. the global moments: contributions added in rank order -- the same bits on every rank
    if (tid < 8) {
        double tot = __builtin_nan("");
        if (arrived) {
            tot = __hip_atomic_load(&own->data[par][0][k * 8 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int r = 1; r < a.world; ++r)
                tot = tot + __hip_atomic_load(&own->data[par][r][k * 8 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        j.gmom[(size_t)a.t * j.gstride + tid] = tot;
    }
}

__global__ __launch_bounds__(kBlock) void p2p_exchange_kernel(P2PJob j, P2PArgs a) { exchange_body(j, a, 0); }

__global__ __launch_bounds__(kBlock) void p2p_exchange_multi_kernel(const P2PJob* __restrict__ jobs, P2PArgs a)
{
    exchange_body(jobs[blockIdx.x], a, (int)blockIdx.x);
}

// slot 6 of each of n result vectors (8 doubles each) := this rank's error word as 0 / 1, so that the all-reduce of the
// result sums carries "some rank's exchange gave up" to every rank
__global__ void p2p_stamp_kernel(double* results, int n, const Mailbox* own)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        results[(size_t)i * 8 + 6] = __hip_atomic_load(&own->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.0 : 0.0;
}

}  // namespace

struct P2P {
    int rank = 0, world = 1;
    Mailbox* own = nullptr;
    Mailbox* box[kP2PMaxWorld] = {};
    unsigned long long epoch = 0;
    double deadline_s = 2.0;
    double first_deadline_s = 30.0;  // the first exchange of a call: ranks may still be loading code objects / allocating
    bool next_is_first = true;
    P2PJob* jobs_dev = nullptr;  // kP2PMaxPricings entries
};

size_t p2p_handle_bytes() { return sizeof(hipIpcMemHandle_t); }

int p2p_export(P2P** out, void* handle_out, std::string* err)
{
    static_assert(sizeof(hipIpcMemHandle_t) == kP2PHandleBytes, "handle size");
    P2P* p = new P2P();
    hipError_t e = hipExtMallocWithFlags((void**)&p->own, sizeof(Mailbox), hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(p->own, 0, sizeof(Mailbox));
    if (e == hipSuccess) e = hipMalloc((void**)&p->jobs_dev, sizeof(P2PJob) * kP2PMaxPricings);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p->own);
    if (e != hipSuccess) {
        if (err) *err = std::string("p2p mailbox: ") + hipGetErrorString(e);
        if (p->own) (void)hipFree(p->own);
        if (p->jobs_dev) (void)hipFree(p->jobs_dev);
        delete p;
        (void)hipGetLastError();
        return 3000 + (int)e;
    }
    memcpy(handle_out, &h, sizeof h);
    *out = p;
    return 0;
}

int p2p_connect(P2P* p, int rank, int world, const void* handles, std::string* err)
{
    if (world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world) {
        if (err) *err = "p2p: rank / world out of range (at most 16 ranks)";
        return -4;
    }
    p->rank = rank;
    p->world = world;
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            p->box[r] = p->own;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * sizeof h, sizeof h);
        void* q = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            if (err) *err = std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e);
            (void)hipGetLastError();
            for (int k = 0; k < r; ++k)
                if (k != rank && p->box[k]) (void)hipIpcCloseMemHandle(p->box[k]);
            for (auto& b : p->box) b = nullptr;
            return 3000 + (int)e;
        }
        p->box[r] = (Mailbox*)q;
    }
    return 0;
}

void p2p_destroy(P2P* p)
{
    if (!p) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < p->world; ++r)
        if (r != p->rank && p->box[r]) (void)hipIpcCloseMemHandle(p->box[r]);
    if (p->own) (void)hipFree(p->own);
    if (p->jobs_dev) (void)hipFree(p->jobs_dev);
    delete p;
}

bool p2p_connected(const P2P* p) { return p && p->box[p->rank] != nullptr && p->world >= 1 && p->box[0] != nullptr; }
int p2p_world(const P2P* p) { return p ? p->world : 0; }
void p2p_set_deadline(P2P* p, double seconds, double first_seconds)
{
    if (p && seconds > 0) p->deadline_s = seconds;
    if (p && first_seconds > 0) p->first_deadline_s = first_seconds;
}
void p2p_begin_call(P2P* p) { if (p) p->next_is_first = true; }

static P2PArgs make_args(P2P* p, int t, int njobs)
{
    P2PArgs a;
    for (int r = 0; r < kP2PMaxWorld; ++r) a.box[r] = p->box[r];
    a.rank = p->rank; a.world = p->world; a.t = t; a.njobs = njobs;
    a.epoch = ++p->epoch;
    // Nothing aligns the ranks before a call's first exchange (a first-use code-object load or a multi-GB allocation
    // on one rank can skew them by seconds): that one waits first_deadline_s, every later one deadline_s.
    const double dl = p->next_is_first && p->first_deadline_s > p->deadline_s ? p->first_deadline_s : p->deadline_s;
    p->next_is_first = false;
    a.deadline_ticks = (unsigned long long)(dl * 1e8);
    return a;
}

hipError_t p2p_exchange_step(P2P* p, hipStream_t st, const LsmWorkspace& w, int t, int nblk)
{
    P2PJob j;
    j.part = w.part; j.gmom = w.gmom; j.nblk = nblk; j.pstride = kPStride; j.gstride = w.gstride; j.pad = 0;
    hipLaunchKernelGGL(p2p_exchange_kernel, dim3(1), dim3(kBlock), 0, st, j, make_args(p, t, 1));
    return hipGetLastError();
}

// K pricings: `jobs_host` (part, gmom, nblk, pstride, gstride per pricing) is uploaded when it changes
hipError_t p2p_set_jobs(P2P* p, hipStream_t st, const double* const* part, double* const* gmom, const int* nblk,
                        const int* gstride, int n)
{
    if (n > kP2PMaxPricings) return hipErrorInvalidValue;
    P2PJob h[kP2PMaxPricings];
    for (int k = 0; k < n; ++k) {
        h[k].part = part[k]; h[k].gmom = gmom[k]; h[k].nblk = nblk[k]; h[k].pstride = kPStride; h[k].gstride = gstride[k];
        h[k].pad = 0;
    }
    hipError_t e = hipMemcpyAsync(p->jobs_dev, h, sizeof(P2PJob) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);  // `h` is on this stack
}

hipError_t p2p_exchange_step_multi(P2P* p, hipStream_t st, int K, int t)
{
    hipLaunchKernelGGL(p2p_exchange_multi_kernel, dim3(K), dim3(kBlock), 0, st, (const P2PJob*)p->jobs_dev, make_args(p, t, K));
    return hipGetLastError();
}

hipError_t p2p_stamp_results(P2P* p, hipStream_t st, double* results, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(p2p_stamp_kernel, dim3((n + 63) / 64), dim3(64), 0, st, results, n, (const Mailbox*)p->own);
    return hipGetLastError();
}

// the sticky error word of this rank's mailbox (0 = every exchange so far completed)
hipError_t p2p_error_word(P2P* p, hipStream_t st, unsigned long long* out)
{
    hipError_t e = hipMemcpyAsync(out, &p->own->error, sizeof *out, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

}  // namespace omc
