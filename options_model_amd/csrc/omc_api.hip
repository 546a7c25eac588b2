// omc_api.hip -- the C ABI of libomc.so (include/omc.h): contexts, workspaces, argument
// checks, launch sequencing, HIP-event timing.  No kernel code here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/omc.h"
#include <sched.h>

#include "omc_batch.h"
#include "omc_comm.h"
#include "omc_p2p.h"
#include "omc_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            char buf_[256];                                                                \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                  \
            g_err = buf_;                                                                  \
            return (int)e_ > 0 ? (int)e_ : 999;                                            \
        }                                                                                  \
    } while (0)

// Option "alloc_limit" (omc_set_option; per process, 0 = none): a single buffer of the library may not grow beyond this many
// bytes -- a request above it fails like a hipMalloc that found no room (hipErrorOutOfMemory).  A memory budget for a
// card shared with other tenants, and the way the tests make ONE rank of a job run out of memory.
static size_t g_alloc_limit = 0;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (g_alloc_limit && bytes > g_alloc_limit) {
            g_err = "hipMalloc refused: " + std::to_string(bytes) + " bytes asked for, option alloc_limit is " +
                    std::to_string(g_alloc_limit) + " (out of memory)";
            return (int)hipErrorOutOfMemory;
        }
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return (int)e;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {  // no room for the 12.5 % growth slack: ask for exactly what is needed
            (void)hipGetLastError();
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();  // a failed hipMalloc must not surface at the next launch's hipGetLastError()
            g_err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
            p = nullptr;
            return (int)e;
        }
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

// What the count call of omc_nn_build_rows (data == NULL) leaves for the call with `data` that follows it: the sweep's
// results on the host, the counts / offsets in the context's scratch.  Valid only for the NEXT library call on the context
// (every entry point clears it in bind()), and only for the same arguments.
struct RowsCache {
    bool valid = false;
    const float* S = nullptr;
    int64_t ld = 0, M = 0;
    int N = 0, is_put = 0;
    double K = 0, r = 0, T = 0;
    int64_t R = 0;
    double st[16] = {0};
};

constexpr size_t kVoteBytes = 1024;  // omc_ctx::seq_vote once a communicator / hook is installed (largest use: 40 doubles)

struct omc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf S, sx, tex, ex, D, part, gmom, betas, part1, result, scratch, sweep_args;
    DevBuf bslab, btable, bres, bdisc;  // batched path: problem slab, table, results, discounts
    DevBuf mlp_part, mlp_loss, mlp_wt;  // NN training: gradient partials, epoch loss, transposed connections
    DevBuf mlp_gred, shard;             // sharded NN training: reduced gradient of a step; epoch selection tables
    DevBuf cn_scratch, cn_data, cn_net, cn_cont;  // per-step ContNet flow: set bookkeeping, rows, net + Adam state, values
    std::vector<char> h_table;
    std::vector<double> h_disc, h_bres;
    std::vector<double> hD;
    int D_N = -1;
    double D_r = 0, D_T = 0;
    const double* D_ptr = nullptr;
    double hres[8];
    double* hres_pin = nullptr;  // pinned + mapped: the fused pricing call's last kernel writes its 8 sums here
    double* hres_dev = nullptr;  // device-side address of hres_pin
    double *seq_pin = nullptr, *seq_dev = nullptr;  // omc_price_american_seq: one 8-double slot per pricing
    int seq_cap = 0;
    hipEvent_t ev_seq = nullptr;
    hipEvent_t ev_entry = nullptr;  // bind_in(): orders a context-owned stream after the device's default stream
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // omc_price_american_seq: further event sets (7 each) for the pricings of a sequence that carry their own
    // kernel timings ("seq_event_stride": every k-th pricing; 0 = the first one only)
    std::vector<hipEvent_t> ev_pool;
    int seq_event_stride = 0;
    // omc_price_american_seq, per-step flows: K pricings advanced by one launch per time step
    DevBuf mS, mstate, mtable;
    DevBuf mb_slab, mb_table, mb_bc;      // omc_mlp_train_epoch_batch: per-problem scratch, table, 1 - beta^step tables
    std::vector<double> mb_bc_host;       // [2][cap]: bc1 then bc2
    double mb_beta1 = -1.0, mb_beta2 = -1.0;
    size_t mb_bc_cap = 0;
    char* mtab_pin = nullptr;  // pinned upload ring for the argument tables (one image per batch of K)
    int mtab_slot = 0;
    int seq_step_k = -1;       // -1: default (what fits the Infinity Cache, <= 16), 1: off, k: at most k pricings per launch
    int seq_step_wgs = 0;      // workgroups one launch of the multi-pricing sweep may use (0: one per CU)
    int gbm_vec = 0, heston_vec = 0;
    // antithetic-folded storage of the fused GBM two-pass pricing (omc_lsm_dev.h; option "fold_antithetic": 0 never,
    // 1 = default: pricings of at least kFoldMinPaths paths over all ranks, 2 always): two cK tables (the overlapped
    // sequence has two pricings in flight), each remembered by what it was filled from
    int fold = 1;
    DevBuf foldC;
    struct FoldKey { int N = -1; double c0 = 0, g = 0; } fold_key[2];
    int world = 1;  // ranks whose sums the hook / communicator adds up (equal shards)
    omc_allreduce_fn hook = nullptr;
    void* hook_user = nullptr;
    omc::Comm* comm = nullptr;  // native RCCL communicator (omc_comm_init); takes precedence over the hook
    // direct write-to-all-peers exchange of the per-step moments (omc_p2p_connect): replaces the per-step all-reduce
    omc::P2P* p2p = nullptr;
    int p2p_use = 1;            // option "p2p_exchange": 0 = keep the collective even when connected
    bool p2p_used = false;      // an exchange was enqueued since the last wait
    double p2p_deadline_s = 2.0;  // option "p2p_deadline_ms": how long an exchange waits for a peer's contribution
    double p2p_first_deadline_s = 30.0;  // "p2p_first_deadline_ms": the same for the FIRST exchange of a call
    // omc_price_american_seq across GPUs: the moment all-reduce of pricing k runs on its own stream while the
    // main stream generates the paths of pricing k+1 into the second path buffer
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_moments[2] = {nullptr, nullptr}, ev_reduced[2] = {nullptr, nullptr};
    RowsCache rows_cache;
    DevBuf S2, seq_local, part1b, gmomb, seq_vote;
    int seq_overlap = -1;  // -1: default (on when the communicator has more than one rank), 0 off, 1 on
    bool defer_result_allreduce = false;  // inside omc_price_american_seq: one collective for all result sums
    // captured per-step sweep (N launches + valuation + finalize), replayed for every pricing of the
    // same geometry; its kernels read their arguments from `sweep_args`
    hipGraph_t sweep_graph = nullptr;
    hipGraphExec_t sweep_exec = nullptr;
    int64_t sg_M = -1, sg_ld = -1;
    int sg_N = -1, sg_sem = -1, sg_vec4 = -1, sg_failed = 0;
    const void* sg_args = nullptr;
    std::vector<char> sweep_img;   // last argument image uploaded to sweep_args
    char* sweep_pin = nullptr;     // pinned upload ring
    int sweep_pin_slot = 0;
    int step_graph = -1;           // -1: environment default (off), 0 off, 1 on
    int device_cus = 0;
    bool distributed() const { return comm != nullptr || hook != nullptr; }
};

namespace {

int check_market(double S0, double K, double T, double r)
{
    if (!(S0 > 0) || !(K > 0) || !(T > 0)) return fail(-1, "S0, K, T must be positive.");
    if (!(r >= 0)) return fail(-2, "r must be non-negative.");
    return 0;
}

int check_sizes(int64_t n_paths, int n_steps)
{
    if (n_paths <= 0 || n_steps <= 0)
        return fail(-3, "num_simulations and num_time_steps must be positive integers.");
    if (n_steps > omc::kMaxSteps) return fail(-8, "num_time_steps exceeds the supported maximum (4094).");
    return 0;
}

int check_matrix(const void* S, int64_t ld, int64_t n_paths)
{
    if (!S) return fail(-7, "null path matrix pointer.");
    if (ld < n_paths) return fail(-6, "leading dimension smaller than n_paths.");
    return 0;
}

int bind(omc_ctx* c)
{
    if (!c) return fail(-7, "null context.");
    c->rows_cache.valid = false;  // (omc_nn_build_rows looks at it before it gets here)
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

// Entry of every call that takes BORROWED device pointers.  A context that owns its stream creates it with
// hipStreamNonBlocking, so nothing orders it after work the caller still has in flight on the device's default
// (null) stream -- which is where PyTorch queues the torch.full / torch.empty / copy that produced the pointer.
// Record a marker on the null stream and make the context's stream wait for it: asynchronous, a few microseconds
// of host time, and the caller's producer is then always ahead of the library's consumer.  A context that borrows
// the caller's stream is ordered by that stream itself.  (Work on OTHER caller streams is the caller's to order:
// include/omc.h, "stream ordering".)
int bind_in(omc_ctx* c)
{
    int rc = bind(c);
    if (rc) return rc;
    if (!c->own_stream) return 0;
    if (!c->ev_entry) HIP_TRY(hipEventCreateWithFlags(&c->ev_entry, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_entry, nullptr));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_entry, 0));
    return 0;
}

// Wait for the stream by polling: a pricing lasts well under a millisecond, and the wake-up
// latency of a blocking hipStreamSynchronize is a visible fraction of that.  The spin is bounded
// in TIME (about 2 ms of polling, yielding the core between polls after the first 50 us), then the
// call blocks: several ranks on a small CPU quota must not burn it all in spin loops.
int wait_stream(omc_ctx* c)
{
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0;; ++i) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) HIP_TRY(q);
        if ((i & 15) == 15) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
            if (us > 2000.0) break;
            if (us > 50.0) sched_yield();
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// workspace for the backward induction on M paths x N steps
int prepare_lsm(omc_ctx* c, int64_t M, int N, double r, double T, bool two_pass,
                bool clear_tables, omc::LsmWorkspace* w)
{
    int rc;
    if ((rc = c->sx.ensure(sizeof(float) * (size_t)M))) return rc;
    if ((rc = c->tex.ensure(sizeof(int32_t) * (size_t)M))) return rc;
    if ((rc = c->ex.ensure(sizeof(float) * (size_t)M + 16))) return rc;
    if ((rc = c->D.ensure(sizeof(double) * (size_t)(N + 1)))) return rc;
    if ((rc = c->part.ensure(sizeof(double) * 2 * 8 * omc::kMaxLsmBlocks))) return rc;
    if ((rc = c->gmom.ensure(sizeof(double) * 8 * (size_t)(N + 1)))) return rc;
    if ((rc = c->betas.ensure(sizeof(double) * 4 * (size_t)(N + 1)))) return rc;
    if ((rc = c->result.ensure(sizeof(double) * 8))) return rc;
    w->part1 = nullptr;
    w->part1_tiles = 0;
    if (two_pass) {
        const size_t tiles = omc::lsm_part1_tiles(M);
        if ((rc = c->part1.ensure(sizeof(double) * 8 * (size_t)(N + 1) * tiles))) return rc;
        w->part1 = (double*)c->part1.p;
        w->part1_tiles = (int64_t)tiles;
    }
    w->sx = (float*)c->sx.p;
    w->tex = (int32_t*)c->tex.p;
    w->live = (float*)c->ex.p;
    w->D = (double*)c->D.p;
    w->part = (double*)c->part.p;
    w->gmom = (double*)c->gmom.p;
    w->betas = (double*)c->betas.p;
    w->result = (double*)c->result.p;
    // discount table computed on the host in double (same libm exp as the oracle); it only
    // depends on (N, r, T), so consecutive pricings of one contract reuse the device copy
    if (c->D_N != N || c->D_r != r || c->D_T != T || c->D_ptr != w->D) {
        c->hD.resize((size_t)N + 1);
        const double dt = T / N;
        for (int k = 0; k <= N; ++k) c->hD[(size_t)k] = std::exp(-r * dt * (double)k);
        HIP_TRY(hipMemcpyAsync(w->D, c->hD.data(), sizeof(double) * (size_t)(N + 1),
                               hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // hD is pageable host memory
        c->D_N = N; c->D_r = r; c->D_T = T; c->D_ptr = w->D;
    }
    // gmom rows 1..N-1, betas rows 1..N-1 and result[0..7] are fully written by the kernels of
    // every flow before anything reads them; rows 0 and N are only ever copied out, so they
    // are cleared just when the caller asked for the tables.
    if (clear_tables) {
        HIP_TRY(hipMemsetAsync(w->gmom, 0, sizeof(double) * 8 * (size_t)(N + 1), c->stream));
        HIP_TRY(hipMemsetAsync(w->betas, 0, sizeof(double) * 4 * (size_t)(N + 1), c->stream));
    }
    return 0;
}

// in-place sum over the ranks of `count` device doubles, ordered on the context's stream: the native
// RCCL communicator when there is one (enqueued from here, no host callback), else the caller's hook
int allreduce(omc_ctx* c, double* dptr, int count)
{
    if (c->comm) {
        std::string err;
        const int rc = omc::comm_allreduce_f64(c->comm, dptr, (size_t)count, 0, c->stream, &err);
        if (rc) return fail(rc, err.c_str());
        return 0;
    }
    if (c->hook) {
        if (c->hook(c->hook_user, dptr, count)) return fail(998, "all-reduce hook failed");
    }
    return 0;
}

// the same for `n` host doubles (n <= 8 * rows of `dev`, a device scratch of the caller's): up, all-reduce, down, wait
int allreduce_host(omc_ctx* c, double* dev, double* host, int n)
{
    int rc;
    HIP_TRY(hipMemcpyAsync(dev, host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if ((rc = allreduce(c, dev, n))) return rc;
    HIP_TRY(hipMemcpyAsync(host, dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// the per-step moments travel by direct peer writes instead of a collective: connected, switched on, and the same
// world as the communicator / hook the rest of the exchange uses
bool p2p_active(const omc_ctx* c)
{
    return c->p2p && c->p2p_use && omc::p2p_connected(c->p2p) && c->distributed() && omc::p2p_world(c->p2p) == c->world;
}

bool step_graph_enabled(const omc_ctx* c)
{
    if (c->step_graph >= 0) return c->step_graph != 0;
    // Off unless asked for: at 1M paths x 252 steps the replayed graph and the 254 plain launches take the
    // same time (1.563 vs 1.562 ms: the host was never the limiter), while every new geometry costs a
    // capture + instantiate of several milliseconds -- a curve whose points differ in step count would
    // pay that per point.
    static const int env = [] {
        const char* e = getenv("OMC_STEP_GRAPH");
        return e ? atoi(e) : 0;
    }();
    return env != 0;
}

void drop_sweep_graph(omc_ctx* c)
{
    if (c->sweep_exec) (void)hipGraphExecDestroy(c->sweep_exec);
    if (c->sweep_graph) (void)hipGraphDestroy(c->sweep_graph);
    c->sweep_exec = nullptr;
    c->sweep_graph = nullptr;
    c->sg_M = -1;
}

// The per-step sweep as ONE graph launch.  Returns 0 when the sweep was enqueued, kNoGraph when the
// caller should launch the kernels one by one (capture unavailable), else an error code.
constexpr int kNoGraph = -12345;
int enqueue_sweep_graph(omc_ctx* c, const omc::LsmProblem& p, const omc::LsmWorkspace& w, int semantics,
                        bool fill_state)
{
    if (c->sg_failed) return kNoGraph;
    const size_t nb = omc::lsm_sweep_args_bytes();
    constexpr int kSlots = 32;
    if (c->sweep_args.ensure(nb)) return kNoGraph;
    if (!c->sweep_pin && hipHostMalloc((void**)&c->sweep_pin, nb * kSlots, hipHostMallocDefault) != hipSuccess) {
        c->sweep_pin = nullptr;
        c->sg_failed = 1;
        (void)hipGetLastError();
        return kNoGraph;
    }
    std::vector<char> img(nb);
    omc::lsm_sweep_args_image(p, w, semantics, fill_state, img.data());
    if (img != c->sweep_img) {
        if (c->sweep_pin_slot == kSlots) {  // the ring wraps: earlier uploads must have been consumed
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->sweep_pin_slot = 0;
        }
        char* slot = c->sweep_pin + nb * (size_t)c->sweep_pin_slot++;
        memcpy(slot, img.data(), nb);
        HIP_TRY(hipMemcpyAsync(c->sweep_args.p, slot, nb, hipMemcpyHostToDevice, c->stream));
        c->sweep_img.swap(img);
    }
    const int vec4 = ((p.M % 4) == 0 && (p.ld % 4) == 0 && ((uintptr_t)p.S % 16) == 0) ? 1 : 0;
    if (!c->sweep_exec || c->sg_M != p.M || c->sg_N != p.N || c->sg_sem != semantics || c->sg_vec4 != vec4 ||
        c->sg_ld != p.ld || c->sg_args != c->sweep_args.p) {
        drop_sweep_graph(c);
        bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const hipError_t le = omc::lsm_sweep_indirect(c->stream, p, w, semantics, c->sweep_args.p);
            const hipError_t ee = hipStreamEndCapture(c->stream, &c->sweep_graph);
            ok = le == hipSuccess && ee == hipSuccess && c->sweep_graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&c->sweep_exec, c->sweep_graph, nullptr, nullptr, 0) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            drop_sweep_graph(c);
            c->sg_failed = 1;
            return kNoGraph;
        }
        c->sg_M = p.M; c->sg_N = p.N; c->sg_sem = semantics; c->sg_vec4 = vec4; c->sg_ld = p.ld;
        c->sg_args = c->sweep_args.p;
    }
    HIP_TRY(hipGraphLaunch(c->sweep_exec, c->stream));
    return 0;
}

// enqueue the whole backward induction on c->stream; sums land in w.result
int enqueue_lsm(omc_ctx* c, const omc::LsmProblem& p, const omc::LsmWorkspace& w, int semantics,
                bool write_state)
{
    hipStream_t st = c->stream;
    int rc;
    if (semantics == OMC_SEM_TWO_PASS) {
        HIP_TRY(omc::lsm_pass1_moments(st, p, w));
        if (c->distributed() && (rc = allreduce(c, w.gmom, 8 * (p.N + 1)))) return rc;
        HIP_TRY(omc::lsm_pass2_apply(st, p, w, write_state, true));  // solves the fits itself
    } else {
        const bool ext = c->distributed();
        const bool flags = semantics == OMC_SEM_REFERENCE;
        int graphed = kNoGraph;
        if (!ext && step_graph_enabled(c)) {
            graphed = enqueue_sweep_graph(c, p, w, semantics, write_state);
            if (graphed != 0 && graphed != kNoGraph) return graphed;
        }
        if (graphed == kNoGraph) {
            const int nblk = omc::lsm_sweep_blocks(p.M);
            if (ext && p2p_active(c)) omc::p2p_begin_call(c->p2p);  // its first exchange absorbs start-up skew
            for (int t = p.N; t >= 1; --t) {
                HIP_TRY(omc::lsm_step(st, p, w, semantics, t, ext));
                if (ext && t >= 2) {
                    if (p2p_active(c)) {  // reduce + publish to all peers + gather + rank-ordered sum: one launch
                        HIP_TRY(omc::p2p_exchange_step(c->p2p, st, w, t - 1, nblk));
                        c->p2p_used = true;
                    } else {
                        HIP_TRY(omc::lsm_reduce_step_moments(st, w, t - 1, nblk));
                        if ((rc = allreduce(c, w.gmom + (size_t)(t - 1) * 8, 8))) return rc;
                    }
                }
            }
            HIP_TRY(omc::lsm_final_reduce(st, p, w, semantics == OMC_SEM_TEXTBOOK ? 0 : 1, flags, write_state));
        }
    }
    // {sum, sumsq, n_exercised, n_zero, sum_nitm, ..} -> global sums.  Slot 4 is built from the moment
    // table, which is ALREADY global on every rank: fill_result divides it by the world size again.
    // (slot 6: "a direct exchange gave up on this rank" -- after the all-reduce every rank knows, check_p2p)
    if (c->p2p_used) HIP_TRY(omc::p2p_stamp_results(c->p2p, st, w.result, 1));
    if (c->distributed() && !c->defer_result_allreduce && (rc = allreduce(c, w.result, 8))) return rc;
    return 0;
}

// after a wait: did a direct exchange give up (its bounded poll ran out) -- on this rank (its sticky error word; the
// sums are NaN then) or on ANY rank (slot 6 of the all-reduced result sums of the n pricings just waited for)?
// Every rank of the job returns the error, not only the one whose deadline ran out.
int check_p2p(omc_ctx* c, const double* h = nullptr, int n = 0)
{
    if (!c->p2p_used) return 0;
    c->p2p_used = false;
    unsigned long long w = 0;
    HIP_TRY(omc::p2p_error_word(c->p2p, c->stream, &w));
    bool peer = false;
    for (int i = 0; h && i < n; ++i) peer = peer || h[(size_t)i * 8 + 6] != 0.0;
    if (w == 1)
        return fail(3100, "direct peer exchange of the per-step moments timed out (a peer's contribution never arrived); "
                          "omc_p2p_disconnect and use the collective");
    if (w || peer)
        return fail(3100, "direct peer exchange of the per-step moments: another rank gave up on an exchange (its deadline "
                          "ran out), so this job's sums are not to be trusted; omc_p2p_disconnect and use the collective");
    return 0;
}

void fill_result(omc_result* res, const double* h, int64_t M, int world = 1)
{
    res->sum = h[0];
    res->sumsq = h[1];
    res->n_paths = M;
    res->n_exercised = (int64_t)llround(h[2]);
    res->n_zero = (int64_t)llround(h[3]);
    res->sum_nitm = (int64_t)llround(h[4] / (double)world);
    res->price = h[0] / (double)M;
    const double var = h[1] / (double)M - res->price * res->price;
    res->std = var > 0 ? std::sqrt(var) : 0.0;
    res->zero_prob = (double)res->n_zero / (double)M;
}

int copy_outputs(omc_ctx* c, const omc::LsmWorkspace& w, int64_t M, int N, double* betas_out,
                 float* sx_out, int32_t* tex_out)
{
    if (betas_out)
        HIP_TRY(hipMemcpyAsync(betas_out, w.betas, sizeof(double) * 4 * (size_t)(N + 1),
                               hipMemcpyDeviceToHost, c->stream));
    if (sx_out)
        HIP_TRY(hipMemcpyAsync(sx_out, w.sx, sizeof(float) * (size_t)M, hipMemcpyDeviceToHost, c->stream));
    if (tex_out)
        HIP_TRY(hipMemcpyAsync(tex_out, w.tex, sizeof(int32_t) * (size_t)M, hipMemcpyDeviceToHost,
                               c->stream));
    return 0;
}

}  // namespace

extern "C" {

int omc_abi_version(void) { return OMC_ABI_VERSION; }

const char* omc_last_error(void) { return g_err.c_str(); }

int omc_device_count(int* count)
{
    if (!count) return fail(-7, "null pointer.");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        g_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return (int)e;
    }
    *count = n;
    return 0;
}

int omc_ctx_create(int device, void* hip_stream, omc_ctx** out)
{
    if (!out) return fail(-7, "null pointer.");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(-4, "no such HIP device.");
    HIP_TRY(hipSetDevice(device));
    omc_ctx* c = new omc_ctx();
    c->device = device;
    auto undo = [&](hipError_t e) {  // release what was created so far, report e
        for (auto& ev : c->ev)
            if (ev) (void)hipEventDestroy(ev);
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return e;
    };
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        const hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            c->stream = nullptr;
            HIP_TRY(undo(e));
        }
        c->own_stream = true;
    }
    for (auto& ev : c->ev) {
        const hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) {
            ev = nullptr;
            HIP_TRY(undo(e));
        }
    }
    if (hipHostMalloc((void**)&c->hres_pin, sizeof(double) * 8, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->hres_dev, c->hres_pin, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (c->hres_pin) (void)hipHostFree(c->hres_pin);
        c->hres_pin = c->hres_dev = nullptr;  // fall back to a device buffer + copy into the pageable member
    }
    (void)hipDeviceGetAttribute(&c->device_cus, hipDeviceAttributeMultiprocessorCount, device);
    *out = c;
    return 0;
}

int omc_ctx_destroy(omc_ctx* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_sweep_graph(c);
    if (c->comm) omc::comm_destroy(c->comm);
    c->comm = nullptr;
    if (c->p2p) omc::p2p_destroy(c->p2p);
    c->p2p = nullptr;
    for (DevBuf* b : {&c->S, &c->sx, &c->tex, &c->ex, &c->D, &c->part, &c->gmom, &c->betas, &c->part1,
                      &c->result, &c->scratch, &c->sweep_args, &c->bslab, &c->btable, &c->bres, &c->bdisc,
                      &c->mlp_part, &c->mlp_loss, &c->mlp_wt, &c->mlp_gred, &c->shard, &c->S2, &c->seq_local, &c->part1b, &c->gmomb, &c->seq_vote, &c->cn_scratch, &c->cn_data, &c->cn_net, &c->cn_cont,
                      &c->mS, &c->mstate, &c->mtable, &c->mb_slab, &c->mb_table, &c->mb_bc})
        b->release();
    if (c->sweep_pin) (void)hipHostFree(c->sweep_pin);
    if (c->mtab_pin) (void)hipHostFree(c->mtab_pin);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    for (int b = 0; b < 2; ++b) {
        if (c->ev_moments[b]) (void)hipEventDestroy(c->ev_moments[b]);
        if (c->ev_reduced[b]) (void)hipEventDestroy(c->ev_reduced[b]);
    }
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->ev_pool)
        if (ev) (void)hipEventDestroy(ev);
    if (c->hres_pin) (void)hipHostFree(c->hres_pin);
    if (c->seq_pin) (void)hipHostFree(c->seq_pin);
    if (c->ev_seq) (void)hipEventDestroy(c->ev_seq);
    if (c->ev_entry) (void)hipEventDestroy(c->ev_entry);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int omc_ctx_device_info(omc_ctx* c, int* device, char* pci_bus_id, int pci_len, char* name, int name_len)
{
    int rc = bind(c);
    if (rc) return rc;
    if (device) *device = c->device;
    if (pci_bus_id && pci_len > 0) {
        pci_bus_id[0] = 0;
        HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, pci_len, c->device));
    }
    if (name && name_len > 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        snprintf(name, (size_t)name_len, "%s", prop.name);
    }
    return 0;
}

int omc_sync(omc_ctx* c)
{
    int rc = bind(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_alloc(omc_ctx* c, size_t bytes, void** dptr)
{
    int rc = bind(c);
    if (rc) return rc;
    if (!dptr) return fail(-7, "null pointer.");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return 0;
}

int omc_free(omc_ctx* c, void* dptr)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (dptr) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(dptr));
    }
    return 0;
}

int omc_memcpy_h2d(omc_ctx* c, void* dst, const void* src, size_t bytes)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (!bytes) return 0;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_memcpy_d2h(omc_ctx* c, void* dst, const void* src, size_t bytes)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (!bytes) return 0;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_set_option(omc_ctx* c, const char* key, int64_t value)
{
    if (!c || !key) return fail(-7, "null pointer.");
    if (!strcmp(key, "gbm_vec")) c->gbm_vec = (int)value;
    else if (!strcmp(key, "alloc_limit")) g_alloc_limit = value > 0 ? (size_t)value : 0;
    else if (!strcmp(key, "fold_antithetic")) c->fold = value <= 0 ? 0 : (value >= 2 ? 2 : 1);
    else if (!strcmp(key, "heston_vec")) c->heston_vec = (int)value;
    else if (!strcmp(key, "world_size")) c->world = value > 0 ? (int)value : 1;
    else if (!strcmp(key, "step_graph")) c->step_graph = value < 0 ? -1 : (value ? 1 : 0);
    else if (!strcmp(key, "seq_overlap")) c->seq_overlap = value < 0 ? -1 : (value ? 1 : 0);
    else if (!strcmp(key, "seq_event_stride")) c->seq_event_stride = value > 0 ? (int)value : 0;
    else if (!strcmp(key, "seq_step_k")) c->seq_step_k = value < 0 ? -1 : (int)(value > 32 ? 32 : value);
    else if (!strcmp(key, "seq_step_wgs")) c->seq_step_wgs = value > 0 ? (int)value : 0;
    else if (!strcmp(key, "p2p_exchange")) c->p2p_use = value ? 1 : 0;
    else if (!strcmp(key, "p2p_deadline_ms") || !strcmp(key, "p2p_first_deadline_ms")) {
        if (value <= 0) return fail(-4, "a p2p deadline must be positive.");
        (key[4] == 'f' ? c->p2p_first_deadline_s : c->p2p_deadline_s) = (double)value * 1e-3;
        if (c->p2p) omc::p2p_set_deadline(c->p2p, c->p2p_deadline_s, c->p2p_first_deadline_s);
    }
    else return fail(-4, "unknown option key.");
    return 0;
}

int omc_set_allreduce_hook(omc_ctx* c, omc_allreduce_fn fn, void* user)
{
    if (!c) return fail(-7, "null context.");
    if (fn) {
        // the few hundred bytes every collective vote / flag exchange goes through: allocated HERE, so that no allocation
        // -- nothing that can fail on one rank alone -- stands between a rank and a collective its peers have entered
        int rc;
        if ((rc = bind(c))) return rc;
        if ((rc = c->seq_vote.ensure(kVoteBytes))) return rc;
    }
    c->hook = fn;
    c->hook_user = user;
    return 0;
}

// ------------------------------------------------------------------ native RCCL communicator
int omc_comm_unique_id(void* uid_out, size_t bytes)
{
    if (!uid_out || bytes < (size_t)omc::kCommUidBytes) return fail(-7, "unique-id buffer must hold 128 bytes.");
    std::string err;
    const int rc = omc::comm_unique_id(uid_out, &err);
    if (rc) return fail(rc, err.c_str());
    return 0;
}

int omc_comm_init(omc_ctx* c, int rank, int world, const void* uid, size_t bytes)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!uid || bytes < (size_t)omc::kCommUidBytes) return fail(-7, "unique id must be 128 bytes.");
    if (world < 1 || rank < 0 || rank >= world) return fail(-4, "rank / world size out of range.");
    if (c->comm) return fail(-4, "this context already has a communicator.");
    std::string err;
    omc::Comm* comm = nullptr;
    if ((rc = c->seq_vote.ensure(kVoteBytes))) return rc;  // (see omc_set_allreduce_hook; before the collective bring-up)
    rc = omc::comm_create(rank, world, uid, &comm, &err);
    if (rc) return fail(rc, err.c_str());
    c->comm = comm;
    c->world = omc::comm_world(comm);
    return 0;
}

int omc_comm_destroy(omc_ctx* c)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!c->comm) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    omc::comm_destroy(c->comm);
    c->comm = nullptr;
    c->world = 1;
    return 0;
}

int omc_comm_info(omc_ctx* c, int* rank, int* world)
{
    if (!c) return fail(-7, "null context.");
    if (rank) *rank = omc::comm_rank(c->comm);
    if (world) *world = c->comm ? omc::comm_world(c->comm) : 0;
    return 0;
}

int omc_comm_allreduce_f64(omc_ctx* c, double* host_inout, int count, int op)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!c->comm) return fail(-4, "no communicator on this context (omc_comm_init).");
    if (!host_inout || count <= 0 || count > 4096) return fail(-7, "bad buffer (1..4096 doubles).");
    if (op != 0 && op != 1) return fail(-4, "op must be 0 (sum) or 1 (max).");
    if ((rc = c->scratch.ensure(sizeof(double) * 4096))) return rc;
    double* d = (double*)c->scratch.p;
    HIP_TRY(hipMemcpyAsync(d, host_inout, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    std::string err;
    rc = omc::comm_allreduce_f64(c->comm, d, (size_t)count, op, c->stream, &err);
    if (rc) return fail(rc, err.c_str());
    HIP_TRY(hipMemcpyAsync(host_inout, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ direct peer exchange (per-step flows)
int omc_p2p_export(omc_ctx* c, void* handle_out, size_t bytes)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!handle_out || bytes < (size_t)omc::kP2PHandleBytes) return fail(-7, "handle buffer must hold 64 bytes.");
    if (c->p2p) return fail(-4, "this context already has a mailbox (omc_p2p_disconnect first).");
    std::string err;
    omc::P2P* p = nullptr;
    rc = omc::p2p_export(&p, handle_out, &err);
    if (rc) return fail(rc, err.c_str());
    c->p2p = p;
    omc::p2p_set_deadline(p, c->p2p_deadline_s, c->p2p_first_deadline_s);
    return 0;
}

int omc_p2p_connect(omc_ctx* c, int rank, int world, const void* handles, size_t bytes)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!c->p2p) return fail(-4, "no mailbox on this context (omc_p2p_export first).");
    if (!handles || world < 1 || bytes < (size_t)world * omc::kP2PHandleBytes)
        return fail(-7, "handles must hold world x 64 bytes, in rank order.");
    std::string err;
    rc = omc::p2p_connect(c->p2p, rank, world, handles, &err);
    if (rc) return fail(rc, err.c_str());
    return 0;
}

int omc_p2p_disconnect(omc_ctx* c)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!c->p2p) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    omc::p2p_destroy(c->p2p);
    c->p2p = nullptr;
    c->p2p_used = false;
    return 0;
}

int omc_p2p_status(omc_ctx* c, int* connected, int* world, uint64_t* error_word)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (connected) *connected = (c->p2p && omc::p2p_connected(c->p2p)) ? 1 : 0;
    if (world) *world = c->p2p ? omc::p2p_world(c->p2p) : 0;
    if (error_word) {
        *error_word = 0;
        if (c->p2p && omc::p2p_connected(c->p2p)) {
            unsigned long long w = 0;
            HIP_TRY(omc::p2p_error_word(c->p2p, c->stream, &w));
            *error_word = w;
        }
    }
    return 0;
}

// ------------------------------------------------------------------ path generation
int omc_gbm_paths_f32(omc_ctx* c, float* S, int64_t ld, int64_t n_paths, int n_steps, double S0,
                      double r, double sigma, double T, uint64_t seed, uint64_t stream,
                      uint64_t pair_offset, int antithetic)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!(S0 > 0) || !(T > 0)) return fail(-1, "S0, K, T must be positive.");
    if (!(sigma > 0)) return fail(-5, "S0, K, T, and sigma must be positive.");
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (antithetic && (n_paths & 1)) return fail(-3, "antithetic layout needs an even n_paths.");
    HIP_TRY(omc::launch_gbm_paths(c->stream, S, ld, n_paths, n_steps, S0, r, sigma, T, seed,
                                  (uint32_t)stream, pair_offset, antithetic, c->gbm_vec));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_heston_paths_f32(omc_ctx* c, float* S, int64_t ld, int64_t n_paths, int n_steps, double S0,
                         double r, double T, double v0, double kappa, double theta, double xi,
                         double rho, uint64_t seed, uint64_t stream, uint64_t pair_offset,
                         int scheme)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!(S0 > 0) || !(T > 0)) return fail(-1, "S0, K, T must be positive.");
    if (!(rho >= -1.0 && rho <= 1.0) || !(v0 >= 0)) return fail(-5, "invalid Heston parameters.");
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (n_paths & 1) return fail(-3, "antithetic layout needs an even n_paths.");
    if (scheme < 0 || scheme > 2) return fail(-4, "unknown Heston scheme.");
    HIP_TRY(omc::launch_heston_paths(c->stream, S, ld, n_paths, n_steps, S0, r, T, v0, kappa, theta,
                                     xi, rho, seed, (uint32_t)stream, pair_offset, scheme,
                                     c->heston_vec));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_gbm_paths_from_normals_f32(omc_ctx* c, float* S, int64_t ld, int64_t n_paths, int n_steps,
                                   double S0, double r, double sigma, double T, const float* Z,
                                   int64_t ldz, int antithetic)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!(S0 > 0) || !(T > 0)) return fail(-1, "S0, K, T must be positive.");
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (!Z) return fail(-7, "null normals pointer.");
    if (antithetic && (n_paths & 1)) return fail(-3, "antithetic layout needs an even n_paths.");
    if (ldz < (antithetic ? n_paths / 2 : n_paths)) return fail(-6, "ldz too small.");
    HIP_TRY(omc::launch_gbm_from_normals(c->stream, S, ld, n_paths, n_steps, S0, r, sigma, T, Z, ldz,
                                         antithetic));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_heston_paths_from_normals_f32(omc_ctx* c, float* S, int64_t ld, int64_t n_paths,
                                      int n_steps, double S0, double r, double T, double v0,
                                      double kappa, double theta, double xi, double rho,
                                      const float* Z1, const float* Z2, int64_t ldz, int scheme)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!(S0 > 0) || !(T > 0)) return fail(-1, "S0, K, T must be positive.");
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (!Z1 || !Z2) return fail(-7, "null normals pointer.");
    if (n_paths & 1) return fail(-3, "antithetic layout needs an even n_paths.");
    if (ldz < n_paths / 2) return fail(-6, "ldz too small.");
    if (scheme < 0 || scheme > 2) return fail(-4, "unknown Heston scheme.");
    HIP_TRY(omc::launch_heston_from_normals(c->stream, S, ld, n_paths, n_steps, S0, r, T, v0, kappa,
                                            theta, xi, rho, Z1, Z2, ldz, scheme));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_philox4x32_10(omc_ctx* c, const uint32_t* in, uint32_t* out, int n)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!in || !out || n <= 0) return fail(-7, "bad arguments.");
    if ((rc = c->scratch.ensure(sizeof(uint32_t) * 10 * (size_t)n))) return rc;
    uint32_t* din = (uint32_t*)c->scratch.p;
    uint32_t* dout = din + 6 * (size_t)n;
    HIP_TRY(hipMemcpyAsync(din, in, sizeof(uint32_t) * 6 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(omc::launch_philox_kat(c->stream, din, dout, n));
    HIP_TRY(hipMemcpyAsync(out, dout, sizeof(uint32_t) * 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_gbm_normals_f32(omc_ctx* c, float* Z, int64_t ldz, int64_t n_pairs, int n_steps,
                        uint64_t seed, uint64_t stream, uint64_t pair_offset)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!Z || n_pairs <= 0 || n_steps <= 0 || ldz < n_pairs) return fail(-7, "bad arguments.");
    HIP_TRY(omc::launch_gbm_normals(c->stream, Z, ldz, n_pairs, n_steps, seed, (uint32_t)stream,
                                    pair_offset));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ backward induction
int omc_lsm_poly(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                 double r, double T, int is_put, int semantics, omc_result* res, double* betas_out,
                 float* sx_out, int32_t* tex_out)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (semantics < 0 || semantics > 2) return fail(-4, "unknown semantics.");
    if (!res) return fail(-7, "null result pointer.");
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, n_paths, n_steps, r, T, semantics == OMC_SEM_TWO_PASS, betas_out != nullptr, &w))) return rc;
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if ((rc = enqueue_lsm(c, p, w, semantics, sx_out || tex_out))) return rc;
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = copy_outputs(c, w, n_paths, n_steps, betas_out, sx_out, tex_out))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((rc = check_p2p(c, c->hres, 1))) return rc;
    memset(res, 0, sizeof *res);
    fill_result(res, c->hres, c->distributed() ? n_paths * c->world : n_paths, c->distributed() ? c->world : 1);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    res->ms_lsm = ms;
    res->ms_total = ms;
    return 0;
}

int omc_lsm_apply_frozen(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps,
                         double K, double r, double T, int is_put, const double* betas,
                         omc_result* res, float* sx_out, int32_t* tex_out)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (!betas || !res) return fail(-7, "null pointer.");
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, n_paths, n_steps, r, T, false, true, &w))) return rc;
    HIP_TRY(hipMemcpyAsync(w.betas, betas, sizeof(double) * 4 * (size_t)(n_steps + 1),
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // `betas` is caller memory
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    HIP_TRY(omc::lsm_pass2_apply(c->stream, p, w, sx_out || tex_out));
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = copy_outputs(c, w, n_paths, n_steps, nullptr, sx_out, tex_out))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(res, 0, sizeof *res);
    fill_result(res, c->hres, n_paths);
    return 0;
}

// The per-step sweep with EXTERNALLY supplied continuation values: cont is a device float32 matrix
// [n_steps+1][ldc]; at step t an in-the-money path that may still exercise does so iff
// payoff > cont[t][j] (strict).  Everything else -- sticky mask or textbook overwrite, discounting,
// valuation time, the returned statistics -- is the code path of omc_lsm_poly's per-step flows.
int omc_lsm_apply_values(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                         double r, double T, int is_put, int semantics, const float* cont, int64_t ldc,
                         omc_result* res, float* sx_out, int32_t* tex_out)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (semantics != OMC_SEM_REFERENCE && semantics != OMC_SEM_TEXTBOOK)
        return fail(-4, "continuation values drive the per-step flows only (semantics 0 or 1).");
    if (!cont || !res) return fail(-7, "null pointer.");
    if (ldc < n_paths) return fail(-6, "ldc smaller than n_paths.");
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, n_paths, n_steps, r, T, false, true, &w))) return rc;
    w.cont = cont;
    w.ldc = ldc;
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    for (int t = n_steps; t >= 1; --t) HIP_TRY(omc::lsm_step(c->stream, p, w, semantics, t, false));
    HIP_TRY(omc::lsm_final_reduce(c->stream, p, w, semantics == OMC_SEM_TEXTBOOK ? 0 : 1,
                                  semantics == OMC_SEM_REFERENCE, sx_out || tex_out));
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = copy_outputs(c, w, n_paths, n_steps, nullptr, sx_out, tex_out))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(res, 0, sizeof *res);
    c->hres[4] = 0.0;  // no regression sets in this mode
    fill_result(res, c->hres, n_paths);
    return 0;
}

// ------------------------------------------------------------------ fused pricing
static int check_params(const omc_params* p)
{
    int rc;
    if (!p) return fail(-7, "null params.");
    if ((rc = check_market(p->S0, p->K, p->T, p->r))) return rc;
    if ((rc = check_sizes(p->n_paths, p->n_steps))) return rc;
    if (p->model == OMC_MODEL_GBM) {
        if (!(p->sigma > 0)) return fail(-5, "S0, K, T, and sigma must be positive.");
    } else if (p->model == OMC_MODEL_HESTON) {
        if (!(p->rho >= -1.0 && p->rho <= 1.0) || !(p->v0 >= 0)) return fail(-5, "invalid Heston parameters.");
        if (p->heston_scheme < 0 || p->heston_scheme > 2) return fail(-4, "unknown Heston scheme.");
        if (!p->antithetic) return fail(-4, "Heston paths are always antithetic.");
    } else {
        return fail(-4, "unknown model.");
    }
    if (p->semantics < 0 || p->semantics > 2) return fail(-4, "unknown semantics.");
    if (p->antithetic && (p->n_paths & 1)) return fail(-3, "antithetic layout needs an even n_paths.");
    return 0;
}

// `fold`: only the FIRST partner of every antithetic pair is generated (n_paths / 2 columns; the same Philox counters,
// hence the same spots, as the first half of the full matrix)
static int enqueue_paths(omc_ctx* c, const omc_params* p, float* S, int64_t ld, bool fold = false)
{
    if (p->model == OMC_MODEL_GBM)
        HIP_TRY(omc::launch_gbm_paths(c->stream, S, ld, fold ? p->n_paths / 2 : p->n_paths, p->n_steps, p->S0, p->r,
                                      p->sigma, p->T, p->seed, (uint32_t)p->stream, p->pair_offset,
                                      fold ? 0 : p->antithetic, c->gbm_vec));
    else
        HIP_TRY(omc::launch_heston_paths(c->stream, S, ld, p->n_paths, p->n_steps, p->S0, p->r, p->T, p->v0,
                                         p->kappa, p->theta, p->xi, p->rho, p->seed, (uint32_t)p->stream,
                                         p->pair_offset, p->heston_scheme, c->heston_vec));
    return 0;
}

// How the fused pricing (the library owns the path matrix) stores the paths of `p`: antithetic GBM in the two-pass flow
// keeps only the first partner of every pair (omc_lsm_dev.h, "antithetic-folded storage") -- *cK is then table `slot` of
// S0^2 exp(2 drift t) / K, (re)filled on the stream when its inputs changed -- everything else the full matrix (*cK null).
// Small pricings stay on the full matrix: they are bound by launch latency, not by bytes (a curve point of a few thousand
// paths would pay a table refill per point for nothing), and omc_price_american_batch -- which prices such members many
// per launch on full storage -- keeps returning the bits of the single calls.  The rule looks at the JOB's paths (local
// paths x ranks), so a sharded pricing and its one-GPU form use the same storage.
constexpr int64_t kFoldMinPaths = 65536;
static bool fold_applies(const omc_ctx* c, const omc_params* p)
{
    if (!c->fold || p->model != OMC_MODEL_GBM || !p->antithetic || p->semantics != OMC_SEM_TWO_PASS) return false;
    return c->fold >= 2 || p->n_paths * (int64_t)(c->world > 0 ? c->world : 1) >= kFoldMinPaths;
}

static int plan_storage(omc_ctx* c, const omc_params* p, int slot, int64_t* ld, const double** cK)
{
    *cK = nullptr;
    *ld = (p->n_paths + 63) / 64 * 64;
    if (!fold_applies(c, p)) return 0;
    int rc;
    const size_t per = (size_t)omc::kMaxSteps + 2;
    if ((rc = c->foldC.ensure(sizeof(double) * 2 * per))) return rc;
    double* tab = (double*)c->foldC.p + per * (size_t)slot;
    double c0, g;
    omc::gbm_fold_constants(p->S0, p->K, p->r, p->sigma, p->T, p->n_steps, &c0, &g);
    omc_ctx::FoldKey& key = c->fold_key[slot];
    if (key.N != p->n_steps || key.c0 != c0 || key.g != g) {
        HIP_TRY(omc::lsm_fold_table(c->stream, tab, p->n_steps, c0, g));
        key.N = p->n_steps; key.c0 = c0; key.g = g;
    }
    *ld = (p->n_paths / 2 + 63) / 64 * 64;
    *cK = tab;
    return 0;
}

// Enqueue one whole pricing (paths + backward induction) on the context's stream; its 8 result sums
// go to `result_dev` (device-visible memory) or, when null, to the workspace's device buffer, which is
// returned through *result_out.  Events are recorded only when `timed`.
static int enqueue_pricing(omc_ctx* c, const omc_params* p, float* S_keep, int64_t ld, double* result_dev,
                           hipEvent_t* evs, double** result_out)
{
    const bool timed = evs != nullptr;
    int rc;
    const int64_t M = p->n_paths;
    const int N = p->n_steps;
    float* S = S_keep;
    const double* cK = nullptr;
    if (S) {
        if (ld < M) return fail(-6, "leading dimension smaller than n_paths.");
    } else {
        if ((rc = plan_storage(c, p, 0, &ld, &cK))) return rc;
        if ((rc = c->S.ensure(sizeof(float) * (size_t)ld * (size_t)(N + 1)))) return rc;
        S = (float*)c->S.p;
    }
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, M, N, p->r, p->T, p->semantics == OMC_SEM_TWO_PASS, false, &w))) return rc;
    omc::LsmProblem prob{S, ld, M, N, p->is_put ? 1 : 0, p->K, p->r, p->T};
    prob.fold_cK = cK;
    if (timed) {
        // (pass 1 starts where the generator ends: evs[1] is its begin; an event costs ~3 us of dispatch gap)
        w.ev_p1_end = evs[4]; w.ev_p2_begin = evs[5]; w.ev_p2_end = evs[6];
        HIP_TRY(hipEventRecord(evs[0], c->stream));
    }
    if ((rc = enqueue_paths(c, p, S, ld, cK != nullptr))) return rc;
    if (timed) HIP_TRY(hipEventRecord(evs[1], c->stream));
    if (result_dev) w.result = result_dev;
    if ((rc = enqueue_lsm(c, prob, w, p->semantics, false))) return rc;
    if (result_out) *result_out = w.result;
    return 0;
}

static int read_kernel_times(const hipEvent_t* evs, const omc_params* p, omc_result* res, bool has_end = true)
{
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, evs[0], evs[1]));
    res->ms_paths = ms;
    if (has_end) {
        HIP_TRY(hipEventElapsedTime(&ms, evs[1], evs[2]));
        res->ms_lsm = ms;
        res->ms_total = res->ms_paths + res->ms_lsm;
    }
    if (p->semantics == OMC_SEM_TWO_PASS && p->n_steps >= 2) {
        HIP_TRY(hipEventElapsedTime(&ms, evs[1], evs[4]));
        res->ms_pass1 = ms;
        HIP_TRY(hipEventElapsedTime(&ms, evs[5], evs[6]));
        res->ms_pass2 = ms;
    }
    return 0;
}

// The event set of the s-th timed pricing of a sequence: the context's own seven events for s == 0, further
// sets from a pool that grows on demand.
static int sample_events(omc_ctx* c, int s, hipEvent_t** out)
{
    if (s == 0) {
        *out = c->ev;
        return 0;
    }
    const size_t need = 7 * (size_t)s;
    while (c->ev_pool.size() < need) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    *out = c->ev_pool.data() + 7 * (size_t)(s - 1);
    return 0;
}

// is pricing i of a sequence one that carries kernel timings, and if so which sample is it
static inline int seq_sample_index(const omc_ctx* c, int i)
{
    if (i == 0) return 0;
    const int k = c->seq_event_stride;
    return (k > 0 && i % k == 0) ? i / k : -1;
}

int omc_price_american(omc_ctx* c, const omc_params* p, omc_result* res, float* S_keep, int64_t ld)
{
    int rc;
    if ((rc = S_keep ? bind_in(c) : bind(c))) return rc;  // only a caller-provided path matrix is borrowed memory
    if ((rc = check_params(p))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    // Single GPU: the finalize kernel stores its 8 sums straight into host-mapped pinned memory (no copy
    // kernel, no extra dependent launch).  With an all-reduce hook the sums stay in device memory for
    // the collective and are copied afterwards.
    const bool zero_copy = c->hres_dev && !c->distributed();
    double* hres = c->hres_pin ? c->hres_pin : c->hres;
    double* result = nullptr;
    if ((rc = enqueue_pricing(c, p, S_keep, ld, zero_copy ? c->hres_dev : nullptr, c->ev, &result))) return rc;
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    if (!zero_copy)
        HIP_TRY(hipMemcpyAsync(hres, result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;
    if ((rc = check_p2p(c, hres, 1))) return rc;
    memset(res, 0, sizeof *res);
    fill_result(res, hres, c->distributed() ? p->n_paths * c->world : p->n_paths,
                c->distributed() ? c->world : 1);  // distributed: sums are global
    res->folded = (!S_keep && fold_applies(c, p)) ? 1 : 0;
    return read_kernel_times(c->ev, p, res);
}

// ------------------------------------------------------------------ per-step ContNet flow (v1 / v2 regressor)
namespace {

// The reference's per-step loop (Options_model.py:112-151 = options_model_2.py:283-312) on a device path
// matrix: for t = N-1 .. 1 { set = in the money & not exercised; skip if empty; fresh net; `epochs` full-batch
// Adam steps; exercise where payoff > net(input) }.  One host read per step (the set's size, which sizes the
// trainer's launches); everything else is stream-ordered.  Leaves the valuation sums in w.result.
int contnet_sweep(omc_ctx* c, const omc::LsmProblem& p, omc::LsmWorkspace& w, int hidden, int epochs, double lr,
                  uint64_t seed, double* rows_total)
{
    int rc;
    const int H = omc::cn_padded_width(hidden);
    const int np = omc::mlp_train_param_count(H, 2);
    const int64_t M = p.M;
    const int N = p.N;
    if ((rc = c->cn_scratch.ensure(omc::cn_scratch_bytes(M)))) return rc;
    if ((rc = c->cn_net.ensure(sizeof(float) * 3 * (size_t)np))) return rc;
    if ((rc = c->cn_cont.ensure(sizeof(float) * (size_t)M))) return rc;
    if ((rc = c->mlp_wt.ensure(omc::mlp_wt_bytes(H, 2)))) return rc;
    if ((rc = c->mlp_loss.ensure(sizeof(double)))) return rc;
    HIP_TRY(hipMemsetAsync(c->mlp_loss.p, 0, sizeof(double), c->stream));  // the trainer adds its losses here (unused)
    float* net = (float*)c->cn_net.p;
    w.cont = (const float*)c->cn_cont.p;
    w.ldc = 0;  // one row, rewritten every step
    const double* hdr_dev = omc::cn_header(p, c->cn_scratch.p);
    *rows_total = 0.0;
    HIP_TRY(omc::lsm_step(c->stream, p, w, OMC_SEM_REFERENCE, N, false));  // state: nobody has exercised
    for (int t = N - 1; t >= 1; --t) {
        const double Dt = c->hD[(size_t)(N - t)];
        HIP_TRY(omc::cn_count(c->stream, p, w, c->cn_scratch.p, t, Dt));
        double hdr[4];
        HIP_TRY(hipMemcpyAsync(hdr, hdr_dev, sizeof hdr, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int64_t R = (int64_t)hdr[0];
        if (R <= 0) continue;  // `if not np.any(itm): continue`
        *rows_total += (double)R;
        if ((rc = c->cn_data.ensure(sizeof(float) * 8 * (size_t)R))) return rc;
        if ((rc = c->mlp_part.ensure(omc::mlp_partial_bytes(H, 2, R)))) return rc;
        HIP_TRY(omc::cn_rows(c->stream, p, w, c->cn_scratch.p, t, Dt, (float*)c->cn_data.p));
        HIP_TRY(omc::cn_init(c->stream, hidden, t, seed, net, net + np, net + 2 * (size_t)np));
        omc::MlpTrainPlan plan;
        plan.data = (const float*)c->cn_data.p;
        plan.params = net; plan.adam_m = net + np; plan.adam_v = net + 2 * (size_t)np;
        plan.partial = (float*)c->mlp_part.p; plan.loss_acc = (double*)c->mlp_loss.p;
        plan.nrows = R; plan.batch = R; plan.hidden = H; plan.layers = 2;
        plan.wt = (float*)c->mlp_wt.p;
        plan.lr = lr; plan.beta1 = 0.9; plan.beta2 = 0.999; plan.eps = 1e-8;  // optim.Adam defaults
        plan.weight_decay = 0.0; plan.dropout = 0.0; plan.seed = 0; plan.shuffle_key = 0;
        for (int e = 0; e < epochs; ++e) {
            plan.first_step = e;
            plan.wt_current = e > 0;
            HIP_TRY(omc::mlp_train_steps(c->stream, plan));
        }
        HIP_TRY(omc::cn_forward(c->stream, p, w, c->cn_scratch.p, t, Dt, hidden, net, (float*)c->cn_cont.p));
        HIP_TRY(omc::lsm_step(c->stream, p, w, OMC_SEM_REFERENCE, t, false));
    }
    return 0;
}

int check_contnet(omc_ctx* c, int hidden, int epochs, double lr)
{
    if (hidden < 1 || omc::cn_padded_width(hidden) < 0) return fail(-4, "nn_hidden must be in 1 .. 128.");
    if (epochs < 0) return fail(-4, "nn_epochs must be non-negative.");
    if (!(lr > 0.0)) return fail(-4, "nn_lr must be positive.");
    if (c->distributed()) return fail(-10, "the per-step network flow runs on one GPU.");
    return 0;
}

int finish_contnet(omc_ctx* c, const omc::LsmProblem& p, omc::LsmWorkspace& w, double rows_total, omc_result* res,
                   float* sx_out, int32_t* tex_out)
{
    int rc;
    HIP_TRY(omc::lsm_final_reduce(c->stream, p, w, 1, true, sx_out || tex_out));
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = copy_outputs(c, w, p.M, p.N, nullptr, sx_out, tex_out))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(res, 0, sizeof *res);
    c->hres[4] = rows_total;
    fill_result(res, c->hres, p.M);
    return 0;
}

}  // namespace

int omc_lsm_contnet(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, double r,
                    double T, int is_put, int nn_hidden, int nn_epochs, double nn_lr, uint64_t nn_seed,
                    omc_result* res, float* sx_out, int32_t* tex_out)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if ((rc = check_contnet(c, nn_hidden, nn_epochs, nn_lr))) return rc;
    if (!res) return fail(-7, "null pointer.");
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, n_paths, n_steps, r, T, false, false, &w))) return rc;
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    double rows = 0.0;
    if ((rc = contnet_sweep(c, p, w, nn_hidden, nn_epochs, nn_lr, nn_seed, &rows))) return rc;
    return finish_contnet(c, p, w, rows, res, sx_out, tex_out);
}

/* the initial parameters of step t's net, in the trainer's padded layout (host float32 [n]) */
int omc_contnet_init_params(omc_ctx* c, int nn_hidden, int t, uint64_t nn_seed, float* params_out, int n)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    const int H = omc::cn_padded_width(nn_hidden);
    if (nn_hidden < 1 || H < 0) return fail(-4, "nn_hidden must be in 1 .. 128.");
    const int np = omc::mlp_train_param_count(H, 2);
    if (!params_out || n != np) return fail(-7, "params_out must hold the padded net's parameters.");
    if ((rc = c->cn_net.ensure(sizeof(float) * 3 * (size_t)np))) return rc;
    float* net = (float*)c->cn_net.p;
    HIP_TRY(omc::cn_init(c->stream, nn_hidden, t, nn_seed, net, net + np, net + 2 * (size_t)np));
    HIP_TRY(hipMemcpyAsync(params_out, net, sizeof(float) * (size_t)np, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_price_american_contnet(omc_ctx* c, const omc_params* p, int nn_hidden, int nn_epochs, double nn_lr,
                               uint64_t nn_seed, omc_result* res)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_params(p))) return rc;
    if (p->semantics != OMC_SEM_REFERENCE)
        return fail(-4, "the per-step network is the regressor of the reference flow (semantics 0).");
    if ((rc = check_contnet(c, nn_hidden, nn_epochs, nn_lr))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    const int64_t M = p->n_paths;
    const int N = p->n_steps;
    const int64_t ld = (M + 63) / 64 * 64;
    if ((rc = c->S.ensure(sizeof(float) * (size_t)ld * (size_t)(N + 1)))) return rc;
    float* S = (float*)c->S.p;
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, M, N, p->r, p->T, false, false, &w))) return rc;
    omc::LsmProblem prob{S, ld, M, N, p->is_put ? 1 : 0, p->K, p->r, p->T};
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if ((rc = enqueue_paths(c, p, S, ld))) return rc;
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    double rows = 0.0;
    if ((rc = contnet_sweep(c, prob, w, nn_hidden, nn_epochs, nn_lr, nn_seed, &rows))) return rc;
    HIP_TRY(omc::lsm_final_reduce(c->stream, prob, w, 1, true, false));
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(res, 0, sizeof *res);
    c->hres[4] = rows;
    fill_result(res, c->hres, M);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    res->ms_paths = ms;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
    res->ms_lsm = ms;
    res->ms_total = res->ms_paths + res->ms_lsm;
    return 0;
}


// n pricings back to back on the stream with NO host synchronisation in between: pricing i + 1 is
// enqueued while pricing i runs, every pricing's sums land in their own slot of a host-mapped buffer,
// one wait at the end.  Results are those of n omc_price_american calls; kernel times are measured on
// the first pricing, ms_total is the average over the sequence (first launch to last completion).
// omc_price_american_seq across GPUs, two-pass flow, native communicator: per pricing the only exchange the
// decisions wait for is the all-reduce of the [N+1][8] moment table between pass 1 and the solves.  Left on
// the main stream it idles the GPU for a collective's latency once per pricing; here it runs on its own
// stream while the main stream generates the NEXT pricing's paths into a second path buffer AND runs its pass 1
// (second partial / moment buffers), and the 8 result sums of all n pricings are all-reduced once, at the end.
// Kernel adjacency stays that of a single pricing -- pass 1 right behind its generator (it starts with the
// rows that are still in the Infinity Cache), pass 2 behind a pass 1 (measured: a pass 2 right behind a
// generator of ANOTHER buffer pays that generator's write-back, +55 us).  Same kernels, same order of every
// reduction: results are bit-identical to the one-at-a-time path.
static bool seq_overlap_enabled(const omc_ctx* c)
{
    if (c->seq_overlap >= 0) return c->seq_overlap != 0;
    static const int env = [] {
        const char* e = getenv("OMC_SEQ_OVERLAP");
        return e ? atoi(e) : -1;
    }();
    if (env >= 0) return env != 0;
    // default: off.  The mechanism uses one communicator from two streams; callers switch it on once the job has
    // checked, with its real communicator, that the overlapped sequence returns the sequential one's bits
    // (bench.py does so before anything is timed).  With one rank it only costs its event hand-overs
    // (0.606 against 0.591 ms per pricing at C2).
    return false;
}

static bool seq_can_overlap(const omc_ctx* c, const omc_params* p, int n)
{
    if (!c->comm || n < 2 || !seq_overlap_enabled(c)) return false;
    for (int i = 0; i < n; ++i)
        if (p[i].semantics != OMC_SEM_TWO_PASS || p[i].n_paths != p[0].n_paths || p[i].n_steps != p[0].n_steps ||
            p[i].r != p[0].r || p[i].T != p[0].T || p[i].n_steps < 2)
            return false;
    return true;
}

static int enqueue_seq_overlapped(omc_ctx* c, const omc_params* p, int n, double* out_pin)
{
    int rc;
    const int64_t M = p[0].n_paths;
    const int N = p[0].n_steps;
    const int64_t ld = (M + 63) / 64 * 64;
    const size_t sbytes = sizeof(float) * (size_t)ld * (size_t)(N + 1);
    if ((rc = c->S.ensure(sbytes))) return rc;
    if ((rc = c->S2.ensure(sbytes))) return rc;
    if ((rc = c->seq_local.ensure(sizeof(double) * 8 * (size_t)n))) return rc;
    omc::LsmWorkspace w[2];
    if ((rc = prepare_lsm(c, M, N, p[0].r, p[0].T, true, false, &w[0]))) return rc;
    // second set of the buffers a pricing owns between its pass 1 and its solves
    if ((rc = c->part1b.ensure(sizeof(double) * 8 * (size_t)(N + 1) * (size_t)w[0].part1_tiles))) return rc;
    if ((rc = c->gmomb.ensure(sizeof(double) * 8 * (size_t)(N + 1)))) return rc;
    w[1] = w[0];
    w[1].part1 = (double*)c->part1b.p;
    w[1].gmom = (double*)c->gmomb.p;
    if (!c->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        if (!c->ev_moments[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_moments[b], hipEventDisableTiming));
        if (!c->ev_reduced[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_reduced[b], hipEventDisableTiming));
    }
    float* Sb[2] = {(float*)c->S.p, (float*)c->S2.p};
    double* local = (double*)c->seq_local.p;
    // every pricing of the sequence chooses its storage for itself (folded matrices are smaller than `sbytes`); the two
    // pricings in flight use different cK tables
    std::vector<int64_t> ldk((size_t)n, ld);
    std::vector<const double*> cKk((size_t)n, nullptr);
    auto problem = [&](int k) {
        omc::LsmProblem q{Sb[k & 1], ldk[(size_t)k], M, N, p[k].is_put ? 1 : 0, p[k].K, p[k].r, p[k].T};
        q.fold_cK = cKk[(size_t)k];
        return q;
    };
    // paths + pass 1 of pricing k on the main stream, then its moment table's all-reduce on the other one
    auto phase_a = [&](int k) -> int {
        const int b = k & 1;
        omc::LsmWorkspace wk = w[b];
        int r2;
        hipEvent_t* evs = nullptr;  // the first pricing (and every seq_event_stride-th) carries timing events
        const int smp = seq_sample_index(c, k);
        if (smp >= 0) {
            if ((r2 = sample_events(c, smp, &evs))) return r2;
            wk.ev_p1_end = evs[4];
            HIP_TRY(hipEventRecord(evs[0], c->stream));
        }
        if ((r2 = plan_storage(c, &p[k], b, &ldk[(size_t)k], &cKk[(size_t)k]))) return r2;
        if ((r2 = enqueue_paths(c, &p[k], Sb[b], ldk[(size_t)k], cKk[(size_t)k] != nullptr))) return r2;
        if (evs) HIP_TRY(hipEventRecord(evs[1], c->stream));
        HIP_TRY(omc::lsm_pass1_moments(c->stream, problem(k), wk));
        HIP_TRY(hipEventRecord(c->ev_moments[b], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_moments[b], 0));
        std::string err;
        if ((r2 = omc::comm_allreduce_f64(c->comm, wk.gmom, (size_t)(8 * (N + 1)), 0, c->comm_stream, &err)))
            return fail(r2, err.c_str());
        HIP_TRY(hipEventRecord(c->ev_reduced[b], c->comm_stream));
        return 0;
    };
    if ((rc = phase_a(0))) return rc;
    for (int k = 0; k < n; ++k) {
        // pricing k+1's paths and pass 1 run while pricing k's collective is in flight
        if (k + 1 < n && (rc = phase_a(k + 1))) return rc;
        const int b = k & 1;
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_reduced[b], 0));
        omc::LsmWorkspace wk = w[b];
        hipEvent_t* evs = nullptr;
        const int smp = seq_sample_index(c, k);
        if (smp >= 0) {
            if ((rc = sample_events(c, smp, &evs))) return rc;
            wk.ev_p2_begin = evs[5]; wk.ev_p2_end = evs[6];
        }
        wk.result = local + 8 * (size_t)k;
        HIP_TRY(omc::lsm_pass2_apply(c->stream, problem(k), wk, false, true));
        if (evs && smp == 0) HIP_TRY(hipEventRecord(evs[2], c->stream));
    }
    if ((rc = allreduce(c, local, 8 * n))) return rc;  // all result sums in one collective
    HIP_TRY(hipMemcpyAsync(out_pin, local, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// ---- per-step flows: K pricings of one geometry per launch --------------------------------------------------
// One launch of the per-step kernel moves 13 MB at C2 and costs ~6 us, of which ~3.6 us are the launch boundary
// and the cold start of a new kernel: a single pricing is at its latency floor, the chip
// is not.  A sequence of pricings that share (n_paths, n_steps, r, T, semantics) therefore advances K of them
// with every launch: K path matrices, K sets of state / partials / fits, ONE launch boundary per time step; across
// GPUs the K moment vectors of a step travel in ONE all-reduce of 8K doubles.  Per pricing the arithmetic and
// the order of every sum are those of its own launches (lsm_step_body), so res[i] keeps the bits of
// omc_price_american(p[i]).
// What the sequence itself allows: depends on the pricings, the context's settings and the environment only -- never on
// this card's free memory -- so every rank of a job computes the same number (they are handed the same sequence).
static int seq_multi_ideal(const omc_ctx* c, const omc_params* p, int n)
{
    if (n < 2) return 1;
    static const int env_k = getenv("OMC_SEQ_STEP_K") ? atoi(getenv("OMC_SEQ_STEP_K")) : -1;
    int k = c->seq_step_k >= 0 ? c->seq_step_k : env_k;
    if (k < 0) {
        // default: as many pricings as keep one launch's rows and state within ~200 MB (measured, tools/exp_step_k.py:
        // at 1M paths the pricing rate rises up to 16-20 pricings per launch -- 0.66 of the HBM roofline -- and falls
        // beyond 240 MB per launch; 250k-path pricings still gain at 32), at most 32; problems so large that fewer
        // than 4 fit are bandwidth-bound one at a time already (8M paths: 0.62 alone, 0.59 with 4 per launch)
        const double per = (p[0].semantics == OMC_SEM_REFERENCE ? 12.0 : 16.0) * (double)p[0].n_paths;
        k = (int)(2.0e8 / per);
        if (k > 32) k = 32;
        if (k < 4) k = 1;
    }
    if (k < 2) return 1;
    if (p[0].semantics == OMC_SEM_TWO_PASS || p[0].n_steps < 1) return 1;
    if (step_graph_enabled(c)) return 1;
    for (int i = 1; i < n; ++i)
        if (p[i].semantics != p[0].semantics || p[i].n_paths != p[0].n_paths || p[i].n_steps != p[0].n_steps ||
            p[i].r != p[0].r || p[i].T != p[0].T)
            return 1;
    if (k > n) k = n;
    if (k > 32) k = 32;
    return k < 2 ? 1 : k;
}

// What THIS card has room for (rank-dependent).  K path matrices stay resident: bounded by a byte budget -- at most 64 GB
// of the 288 (OMC_SEQ_STEP_BYTES), and never more than 80 % of what is free on this card right now plus what the context
// already holds for them (a card shared with torch or with other ranks has less; seq_multi_reserve also halves K when
// the allocation fails all the same).
static int seq_multi_fit(const omc_ctx* c, const omc_params* p, int k)
{
    static const double cap = getenv("OMC_SEQ_STEP_BYTES") ? atof(getenv("OMC_SEQ_STEP_BYTES")) : 64e9;
    double budget = cap;
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(c->device);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double avail = 0.8 * (double)free_b + (double)c->mS.cap;
        if (avail < budget) budget = avail;
    } else {
        (void)hipGetLastError();
    }
    const int64_t ld = (p[0].n_paths + 63) / 64 * 64;
    const double sbytes = 4.0 * (double)ld * (double)(p[0].n_steps + 1);
    const int fit = (int)(budget / sbytes);
    if (k > fit) k = fit;
    return k < 2 ? 1 : k;
}

// this rank's own estimate (omc_seq_step_width; a job agrees on the smallest in seq_multi_reserve)
static int seq_multi_width(const omc_ctx* c, const omc_params* p, int n)
{
    const int k = seq_multi_ideal(c, p, n);
    return k < 2 ? 1 : seq_multi_fit(c, p, k);
}

// Device memory of the K-pricings-per-launch sweep (K path matrices + per-pricing state).  -> 0, or the HIP error.
static int seq_multi_alloc(omc_ctx* c, int64_t M, int N, int K)
{
    int rc;
    const int64_t ld = (M + 63) / 64 * 64;
    const size_t sbytes = sizeof(float) * (size_t)ld * (size_t)(N + 1);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t per = up(sizeof(float) * (size_t)M) + up(sizeof(int32_t) * (size_t)M) + up(sizeof(float) * (size_t)M + 16) +
                       up(sizeof(double) * 2 * 8 * omc::kMaxLsmBlocks) + up(sizeof(double) * 4 * (size_t)(N + 1));
    const size_t gbytes = up(sizeof(double) * 8 * (size_t)K * (size_t)(N + 1));
    if ((rc = c->mS.ensure(sbytes * (size_t)K))) return rc;
    if ((rc = c->mstate.ensure(gbytes + per * (size_t)K))) return rc;
    return c->mtable.ensure(omc::lsm_sweep_args_bytes() * (size_t)K);
}

// Reserve for K pricings per launch; when the card has no room (shared with torch, several ranks on one device, a
// smaller card) halve K down to one pricing at a time instead of failing the sequence.
// Ranks of one job must agree on K (their per-step collectives carry 8K doubles, their direct exchanges K jobs), and a
// rank must never skip a collective its peers enter.  So: whether a vote takes place depends on seq_multi_ideal alone
// (the same on every rank); when it does, EVERY rank votes -- also one whose own K came out as 1, also one whose
// allocation failed for another reason than memory -- through the context's generic all-reduce (communicator or
// hook): a one-hot vector of 33 counters plus an error counter, summed; the job takes the smallest K anybody voted
// for, and fails everywhere if anybody reported an error.
static int seq_multi_reserve(omc_ctx* c, const omc_params* p, int n, int* K_out)
{
    *K_out = 1;
    const int ideal = seq_multi_ideal(c, p, n);
    if (ideal < 2) return 0;  // (every rank takes this branch together)
    int K = seq_multi_fit(c, p, ideal), rc = 0, err = 0;
    std::string err_text;
    while (K >= 2 && (rc = seq_multi_alloc(c, p[0].n_paths, p[0].n_steps, K)) != 0) {
        if (rc != (int)hipErrorOutOfMemory && rc != (int)hipErrorMemoryAllocation) {
            err = rc;
            err_text = g_err;
            break;
        }
        K /= 2;
    }
    if (K < 2 || err) K = 1;
    if (c->distributed()) {
        constexpr int kVote = 34;  // K = 1 .. 32 one-hot (slot K), slot 33 = ranks in trouble
        double vote[kVote] = {0};
        vote[K] = 1.0;
        vote[33] = err ? 1.0 : 0.0;
        if ((rc = c->seq_vote.ensure(sizeof vote))) return rc;  // (never allocates: the buffer exists since the communicator / hook was installed)
        if ((rc = allreduce_host(c, (double*)c->seq_vote.p, vote, kVote))) return rc;
        if (vote[33] > 0.0) {
            if (err) return fail(err, err_text.c_str());
            return fail(3101, "another rank of the job could not reserve memory for the sequence of pricings.");
        }
        K = 1;
        for (int k = 1; k <= 32; ++k)
            if (vote[k] > 0.0) { K = k; break; }
    } else if (err) {
        return fail(err, err_text.c_str());
    }
    *K_out = K < 2 ? 1 : K;
    return 0;
}

static int enqueue_seq_step_multi(omc_ctx* c, const omc_params* p, int n, int K, double* dst)
{
    int rc;
    const int64_t M = p[0].n_paths;
    const int N = p[0].n_steps;
    const int sem = p[0].semantics;
    const int64_t ld = (M + 63) / 64 * 64;
    const size_t sbytes = sizeof(float) * (size_t)ld * (size_t)(N + 1);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_sx = 0, o_tex = o_sx + up(sizeof(float) * (size_t)M), o_ex = o_tex + up(sizeof(int32_t) * (size_t)M),
                 o_part = o_ex + up(sizeof(float) * (size_t)M + 16), o_betas = o_part + up(sizeof(double) * 2 * 8 * omc::kMaxLsmBlocks),
                 per = o_betas + up(sizeof(double) * 4 * (size_t)(N + 1));
    const size_t gbytes = up(sizeof(double) * 8 * (size_t)K * (size_t)(N + 1));
    if ((rc = c->mS.ensure(sbytes * (size_t)K))) return rc;
    if ((rc = c->mstate.ensure(gbytes + per * (size_t)K))) return rc;
    const size_t eb = omc::lsm_sweep_args_bytes();
    const size_t tbytes = eb * (size_t)K;
    if ((rc = c->mtable.ensure(tbytes))) return rc;
    constexpr int kSlots = 32;
    constexpr size_t kSlotBytes = 32 * 1024;
    if (tbytes > kSlotBytes) return fail(-4, "argument table of the multi-pricing sweep exceeds its upload slot.");
    if (!c->mtab_pin) HIP_TRY(hipHostMalloc((void**)&c->mtab_pin, kSlotBytes * kSlots, hipHostMallocDefault));
    omc::LsmWorkspace w0;
    if ((rc = prepare_lsm(c, M, N, p[0].r, p[0].T, false, false, &w0))) return rc;  // discount table (+ unused singles)
    const bool ext = c->distributed();
    const bool vec4 = (M % 4) == 0;  // ld is a multiple of 64 and every matrix starts 256-byte aligned
    char* state = (char*)c->mstate.p;
    double* gmomK = (double*)state;
    for (int i0 = 0; i0 < n; i0 += K) {
        const int Kb = n - i0 < K ? n - i0 : K;
        const int G = omc::lsm_multi_groups(M, Kb, c->seq_step_wgs > 0 ? c->seq_step_wgs : c->device_cus);
        if (c->mtab_slot == kSlots) {  // the ring wraps: earlier uploads must have been consumed
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->mtab_slot = 0;
        }
        char* img = c->mtab_pin + kSlotBytes * (size_t)c->mtab_slot++;
        for (int k = 0; k < Kb; ++k) {
            const omc_params& q = p[i0 + k];
            char* st = state + gbytes + per * (size_t)k;
            omc::LsmWorkspace w = w0;
            w.sx = (float*)(st + o_sx); w.tex = (int32_t*)(st + o_tex); w.live = (float*)(st + o_ex);
            w.part = (double*)(st + o_part); w.betas = (double*)(st + o_betas);
            w.gmom = gmomK + 8 * (size_t)k; w.gstride = 8 * Kb;
            w.result = dst + 8 * (size_t)(i0 + k);
            omc::LsmProblem prob{(const float*)((char*)c->mS.p + sbytes * (size_t)k), ld, M, N, q.is_put ? 1 : 0, q.K, q.r, q.T};
            omc::lsm_sweep_args_image(prob, w, sem, false, img + eb * (size_t)k, ext);
        }
        HIP_TRY(hipMemcpyAsync(c->mtable.p, img, eb * (size_t)Kb, hipMemcpyHostToDevice, c->stream));
        const bool p2p = ext && p2p_active(c) && Kb <= omc::kP2PMaxPricings;
        if (p2p) {  // where each pricing's partials are and where its global moments go
            const double* parts[omc::kP2PMaxPricings];
            double* gm[omc::kP2PMaxPricings];
            int nb[omc::kP2PMaxPricings], gs[omc::kP2PMaxPricings];
            for (int k = 0; k < Kb; ++k) {
                parts[k] = (const double*)(state + gbytes + per * (size_t)k + o_part);
                gm[k] = gmomK + 8 * (size_t)k;
                nb[k] = omc::lsm_sweep_blocks(M);
                gs[k] = 8 * Kb;
            }
            HIP_TRY(omc::p2p_set_jobs(c->p2p, c->stream, parts, gm, nb, gs, Kb));
        }
        if (i0 == 0 && p2p) omc::p2p_begin_call(c->p2p);  // its first exchange absorbs start-up skew
        if (i0 == 0) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        for (int k = 0; k < Kb; ++k)
            if ((rc = enqueue_paths(c, &p[i0 + k], (float*)((char*)c->mS.p + sbytes * (size_t)k), ld))) return rc;
        if (i0 == 0) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        for (int t = N; t >= 1; --t) {
            HIP_TRY(omc::lsm_step_multi(c->stream, c->mtable.p, Kb, G, sem, vec4, N, t));
            if (ext && t >= 2) {
                if (p2p) {
                    HIP_TRY(omc::p2p_exchange_step_multi(c->p2p, c->stream, Kb, t - 1));
                    c->p2p_used = true;
                } else {
                    HIP_TRY(omc::lsm_reduce_step_moments_multi(c->stream, c->mtable.p, Kb, t - 1));
                    if ((rc = allreduce(c, gmomK + (size_t)(t - 1) * 8 * (size_t)Kb, 8 * Kb))) return rc;  // K fits' moments, one collective
                }
            }
        }
        HIP_TRY(omc::lsm_final_multi(c->stream, c->mtable.p, Kb, M));
        if (i0 == 0) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    }
    return 0;
}

int omc_seq_step_width(omc_ctx* c, const omc_params* p, int n)
{
    if (!c || !p || n <= 0) return 0;
    for (int i = 0; i < n; ++i)
        if (check_params(&p[i])) return 0;
    return seq_multi_width(c, p, n);
}

int omc_price_american_seq(omc_ctx* c, const omc_params* p, int n, omc_result* res)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if (!p || !res || n <= 0) return fail(-7, "null pointer or empty sequence.");
    for (int i = 0; i < n; ++i)
        if ((rc = check_params(&p[i]))) return rc;
    if (!c->hres_dev) {  // no host-mapped memory on this system: one pricing at a time
        for (int i = 0; i < n; ++i)
            if ((rc = omc_price_american(c, &p[i], &res[i], nullptr, 0))) return rc;
        return 0;
    }
    if (c->seq_cap < n) {
        if (c->seq_pin) (void)hipHostFree(c->seq_pin);
        c->seq_pin = c->seq_dev = nullptr;
        c->seq_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&c->seq_pin, sizeof(double) * 8 * (size_t)n, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&c->seq_dev, c->seq_pin, 0));
        c->seq_cap = n;
    }
    hipEvent_t ev_end = c->ev[2];
    if (!c->ev_seq) HIP_TRY(hipEventCreate(&c->ev_seq));
    ev_end = c->ev_seq;
    const bool overlapped = seq_can_overlap(c, p, n);
    int multi = 1;
    if (!overlapped && (rc = seq_multi_reserve(c, p, n, &multi))) return rc;
    if (overlapped && (rc = enqueue_seq_overlapped(c, p, n, c->seq_pin))) return rc;
    // across GPUs the sums stay in device memory (one slot per pricing) and are all-reduced together after
    // the last pricing -- the hook / communicator sees ONE call with 8n doubles -- then copied out
    const bool dist = c->distributed() && !overlapped;
    if (dist && (rc = c->seq_local.ensure(sizeof(double) * 8 * (size_t)n))) return rc;
    double* local = (double*)c->seq_local.p;
    c->defer_result_allreduce = dist;
    if (multi > 1 && (rc = enqueue_seq_step_multi(c, p, n, multi, dist ? local : c->seq_dev))) {
        c->defer_result_allreduce = false;
        return rc;
    }
    for (int i = 0; i < n && !overlapped && multi <= 1; ++i) {
        hipEvent_t* evs = nullptr;
        const int smp = seq_sample_index(c, i);
        if (smp >= 0 && (rc = sample_events(c, smp, &evs))) break;
        rc = enqueue_pricing(c, &p[i], nullptr, 0, dist ? local + 8 * (size_t)i : c->seq_dev + 8 * (size_t)i,
                             evs, nullptr);
        if (rc) break;
        if (evs && smp == 0 && hipEventRecord(evs[2], c->stream) != hipSuccess) { rc = fail(999, "hipEventRecord failed"); break; }
    }
    c->defer_result_allreduce = false;
    if (rc) return rc;
    if (dist) {
        if (c->p2p_used) HIP_TRY(omc::p2p_stamp_results(c->p2p, c->stream, local, n));  // (enqueue_lsm stamps one at a time)
        if ((rc = allreduce(c, local, 8 * n))) return rc;
        HIP_TRY(hipMemcpyAsync(c->seq_pin, local, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipEventRecord(ev_end, c->stream));
    if ((rc = wait_stream(c))) return rc;
    if ((rc = check_p2p(c, c->seq_pin, n))) return rc;
    float ms_all = 0;
    HIP_TRY(hipEventElapsedTime(&ms_all, c->ev[0], ev_end));
    // kernel times: a pricing that carried events reports its own, the others those of the latest one before them
    omc_result timed;
    memset(&timed, 0, sizeof timed);
    for (int i = 0; i < n; ++i) {
        const int smp = seq_sample_index(c, i);
        if (smp >= 0 && (multi <= 1 || i == 0)) {
            hipEvent_t* evs = nullptr;
            if ((rc = sample_events(c, smp, &evs))) return rc;
            if ((rc = read_kernel_times(evs, &p[i], &timed, smp == 0))) return rc;
            if (multi > 1) timed.ms_paths /= (double)multi;  // ev[0]..ev[1] spans the first batch's K generators
        }
        memset(&res[i], 0, sizeof res[i]);
        fill_result(&res[i], c->seq_pin + 8 * (size_t)i, c->distributed() ? p[i].n_paths * c->world : p[i].n_paths,
                    c->distributed() ? c->world : 1);
        res[i].ms_paths = timed.ms_paths;
        res[i].ms_pass1 = timed.ms_pass1;
        res[i].ms_pass2 = timed.ms_pass2;
        res[i].ms_total = ms_all / (float)n;
        res[i].ms_lsm = res[i].ms_total - timed.ms_paths;
        res[i].timed = (smp >= 0 && (multi <= 1 || i == 0)) ? 1 : 0;
        res[i].folded = fold_applies(c, &p[i]) ? 1 : 0;
    }
    return 0;
}

int omc_price_european(omc_ctx* c, const omc_params* p, omc_result* res)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_params(p))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    if ((rc = c->part.ensure(sizeof(double) * 2 * 8 * omc::kMaxLsmBlocks))) return rc;
    if ((rc = c->result.ensure(sizeof(double) * 8))) return rc;
    double* part = (double*)c->part.p;
    HIP_TRY(hipMemsetAsync(part, 0, sizeof(double) * 8 * omc::kMaxLsmBlocks, c->stream));
    int nblk = 0;
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(omc::launch_terminal(c->stream, part, &nblk, p->model, p->heston_scheme, p->antithetic,
                                 p->n_paths, p->n_steps, p->S0, p->K, p->r, p->sigma, p->T, p->v0,
                                 p->kappa, p->theta, p->xi, p->rho, p->is_put ? 1 : 0, p->seed,
                                 (uint32_t)p->stream, p->pair_offset));
    HIP_TRY(omc::lsm_finalize(c->stream, part, nullptr, (double*)c->result.p, nblk, 0));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipMemcpyAsync(c->hres, c->result.p, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(res, 0, sizeof *res);
    c->hres[2] = 0.0;
    c->hres[4] = 0.0;
    fill_result(res, c->hres, p->n_paths);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    res->ms_paths = ms;
    res->ms_total = ms;
    return 0;
}

// ------------------------------------------------------------------ calibrator inner loop
int omc_heston_price_strikes(omc_ctx* c, int64_t n_paths, int n_steps, double S0, double r, double T,
                             double v0, double kappa, double theta, double xi, double rho,
                             uint64_t seed, uint64_t stream, int scheme, const double* strikes,
                             int n_strikes, int is_put, double* prices, double* stderrs)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!(S0 > 0) || !(T > 0)) return fail(-1, "S0, K, T must be positive.");
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if (n_paths & 1) return fail(-3, "antithetic layout needs an even n_paths.");
    if (scheme < 0 || scheme > 2) return fail(-4, "unknown Heston scheme.");
    if (!(rho >= -1.0 && rho <= 1.0) || !(v0 >= 0)) return fail(-5, "invalid Heston parameters.");
    if (!strikes || !prices || n_strikes <= 0) return fail(-7, "bad strike arguments.");
    if (n_paths > (int64_t)65535 * 4096) return fail(-3, "at most 268,431,360 paths per expiry.");
    const size_t st_bytes = sizeof(float) * (size_t)n_paths;
    const size_t k_bytes = sizeof(double) * (size_t)n_strikes;
    if ((rc = c->scratch.ensure(st_bytes + 256 + 3 * k_bytes + omc::payoff_partial_bytes(n_paths, n_strikes)))) return rc;
    float* ST = (float*)c->scratch.p;
    double* Kd = (double*)((char*)c->scratch.p + (st_bytes + 255) / 256 * 256);
    double* out = Kd + n_strikes;
    double* part = out + 2 * (size_t)n_strikes;
    HIP_TRY(hipMemcpyAsync(Kd, strikes, k_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(omc::launch_heston_terminal_store(c->stream, ST, n_paths, n_steps, S0, r, T, v0, kappa, theta,
                                              xi, rho, seed, (uint32_t)stream, 0, scheme));
    HIP_TRY(omc::launch_payoff_means(c->stream, ST, n_paths, Kd, n_strikes, is_put ? 1 : 0, part, out));
    std::vector<double> h(2 * (size_t)n_strikes);
    HIP_TRY(hipMemcpyAsync(h.data(), out, 2 * k_bytes, hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;  // (polling: the call lasts ~0.1 ms)
    const double df = std::exp(-r * T), M = (double)n_paths;
    for (int k = 0; k < n_strikes; ++k) {
        const double mean = h[2 * (size_t)k] / M;
        const double var = h[2 * (size_t)k + 1] / M - mean * mean;
        prices[k] = df * mean;
        if (stderrs) stderrs[k] = df * std::sqrt((var > 0 ? var : 0.0) / M);
    }
    return 0;
}

// A whole quote surface in one launch set: what one evaluation of the calibrator's objective asks for
// (heston_calibration.py:283-312, 404-472: ~60 quotes over a handful of expiries, per optimizer iteration).  Every
// expiry is simulated on its own Philox sub-stream (streams[e]) and every quote averaged over ITS expiry's terminal
// spots -- two launches, one table upload, one read-back, one wait, instead of that per expiry; each quote comes back
// with the bits of its own omc_heston_price_strikes(T = expiries[expiry_of[q]], stream = streams[expiry_of[q]]) call.
int omc_heston_price_surface(omc_ctx* c, int64_t n_paths, int n_steps, double S0, double r, double v0, double kappa,
                             double theta, double xi, double rho, uint64_t seed, int scheme, const double* expiries,
                             const uint64_t* streams, int n_expiries, const double* strikes, const int32_t* expiry_of,
                             int n_quotes, int is_put, double* prices, double* stderrs)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!(S0 > 0)) return fail(-1, "S0, K, T must be positive.");
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if (n_paths & 1) return fail(-3, "antithetic layout needs an even n_paths.");
    if (scheme < 0 || scheme > 2) return fail(-4, "unknown Heston scheme.");
    if (!(rho >= -1.0 && rho <= 1.0) || !(v0 >= 0)) return fail(-5, "invalid Heston parameters.");
    if (!expiries || !streams || n_expiries <= 0 || n_expiries > 65535) return fail(-7, "bad expiry arguments (1 .. 65535 expiries).");
    if (!strikes || !expiry_of || !prices || n_quotes <= 0) return fail(-7, "bad strike arguments.");
    if (n_paths > (int64_t)65535 * 4096) return fail(-3, "at most 268,431,360 paths per expiry.");
    for (int e = 0; e < n_expiries; ++e)
        if (!(expiries[e] > 0)) return fail(-1, "S0, K, T must be positive.");
    for (int q = 0; q < n_quotes; ++q)
        if (expiry_of[q] < 0 || expiry_of[q] >= n_expiries) return fail(-4, "expiry_of[q] must index the expiries.");
    const int64_t ldst = (n_paths + 63) / 64 * 64;
    const size_t st_bytes = (sizeof(float) * (size_t)ldst * (size_t)n_expiries + 255) / 256 * 256;
    // ONE upload per call: [expiry table | strikes | quote -> expiry] as one host image behind the terminal spots
    const size_t tab_bytes = (omc::heston_surface_table_bytes(n_expiries) + 255) / 256 * 256;
    const size_t k_bytes = sizeof(double) * (size_t)n_quotes, e_bytes = (sizeof(int32_t) * (size_t)n_quotes + 255) / 256 * 256;
    const size_t img_bytes = tab_bytes + (k_bytes + 255) / 256 * 256 + e_bytes;
    if ((rc = c->scratch.ensure(st_bytes + img_bytes + 2 * k_bytes + 256 + omc::payoff_partial_bytes(n_paths, n_quotes)))) return rc;
    char* base = (char*)c->scratch.p;
    float* ST = (float*)base;
    char* img_d = base + st_bytes;
    const void* tab = img_d;
    const double* Kd = (const double*)(img_d + tab_bytes);
    const int32_t* eo = (const int32_t*)(img_d + tab_bytes + (k_bytes + 255) / 256 * 256);
    double* out = (double*)(img_d + img_bytes);
    double* part = out + 2 * (size_t)n_quotes;
    // the host image must outlive the asynchronous copy (pageable memory): the context keeps it until the wait below
    c->h_table.resize(img_bytes + sizeof(uint32_t) * (size_t)n_expiries);
    char* img_h = c->h_table.data();
    uint32_t* st32 = (uint32_t*)(img_h + img_bytes);
    for (int e = 0; e < n_expiries; ++e) st32[e] = (uint32_t)streams[e];
    omc::heston_surface_fill_table(img_h, n_steps, r, expiries, st32, n_expiries, kappa, theta, xi, rho);
    memcpy(img_h + tab_bytes, strikes, k_bytes);
    memcpy(img_h + tab_bytes + (k_bytes + 255) / 256 * 256, expiry_of, sizeof(int32_t) * (size_t)n_quotes);
    HIP_TRY(hipMemcpyAsync(img_d, img_h, img_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(omc::launch_heston_terminal_surface(c->stream, ST, ldst, n_paths, n_steps, S0, n_expiries, v0, seed, 0, scheme, tab));
    HIP_TRY(omc::launch_payoff_means_surface(c->stream, ST, ldst, n_paths, Kd, eo, n_quotes, is_put ? 1 : 0, part, out));
    std::vector<double> h(2 * (size_t)n_quotes);
    HIP_TRY(hipMemcpyAsync(h.data(), out, 2 * k_bytes, hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;  // (polling: the call lasts ~0.1 ms)
    const double M = (double)n_paths;
    for (int q = 0; q < n_quotes; ++q) {
        const double df = std::exp(-r * expiries[expiry_of[q]]);
        const double mean = h[2 * (size_t)q] / M;
        const double var = h[2 * (size_t)q + 1] / M - mean * mean;
        prices[q] = df * mean;
        if (stderrs) stderrs[q] = df * std::sqrt((var > 0 ? var : 0.0) / M);
    }
    return 0;
}

// ------------------------------------------------------------------ batched small pricings
static int check_batch(const omc_params* p, int n)
{
    if (!p || n <= 0) return fail(-7, "empty batch.");
    if (n > 65535) return fail(-3, "batch too large (max 65535 problems per call).");
    for (int i = 0; i < n; ++i) {
        int rc = check_params(&p[i]);
        if (rc) return rc;
        if (p[i].model != p[0].model || p[i].semantics != p[0].semantics ||
            p[i].antithetic != p[0].antithetic || p[i].heston_scheme != p[0].heston_scheme)
            return fail(-4, "a batch must share model, semantics, antithetic and Heston scheme.");
    }
    return 0;
}

static int generator_id(const omc_params* p)
{
    if (p->model == OMC_MODEL_GBM) return p->antithetic ? 0 : 1;
    return 2 + p->heston_scheme;  // 2 reference clamp, 3 full truncation, 4 calibrator scheme
}

static int run_batch_group(omc_ctx* c, const omc_params* p, int n, omc_result* res, bool american)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_batch(p, n))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    if (c->distributed()) return fail(-4, "batched pricing is single-GPU (no all-reduce hook / communicator).");
    const bool two_pass = p[0].semantics == OMC_SEM_TWO_PASS;
    const size_t slab = omc::batch_slab_bytes(p, n, american, two_pass);
    const size_t nd = american ? omc::batch_discount_doubles(p, n) : 0;
    if ((rc = c->bslab.ensure(slab))) return rc;
    if ((rc = c->btable.ensure(omc::batch_table_bytes(n)))) return rc;
    if ((rc = c->bres.ensure(sizeof(double) * 8 * (size_t)n))) return rc;
    if ((rc = c->bdisc.ensure(sizeof(double) * (nd ? nd : 1)))) return rc;
    c->h_table.resize(omc::batch_table_bytes(n));
    c->h_disc.resize(nd ? nd : 1);
    c->h_bres.resize(8 * (size_t)n);
    omc::BatchExtents e;
    omc::batch_build(p, n, american, two_pass, (char*)c->bslab.p, (double*)c->bres.p,
                     (double*)c->bdisc.p, c->h_table.data(), c->h_disc.data(), &e);
    HIP_TRY(hipMemcpyAsync(c->btable.p, c->h_table.data(), c->h_table.size(), hipMemcpyHostToDevice,
                           c->stream));
    if (nd)
        HIP_TRY(hipMemcpyAsync(c->bdisc.p, c->h_disc.data(), sizeof(double) * nd, hipMemcpyHostToDevice,
                               c->stream));
    if (!american) HIP_TRY(hipMemsetAsync(c->bslab.p, 0, slab, c->stream));  // unused partial rows
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    const int gen = generator_id(&p[0]);
    if (american) {
        HIP_TRY(omc::batch_paths(c->stream, c->btable.p, n, e, gen));
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        HIP_TRY(omc::batch_lsm(c->stream, c->btable.p, n, e, p[0].semantics));
    } else {
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        HIP_TRY(omc::batch_terminal(c->stream, c->btable.p, n, e, gen));
    }
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_bres.data(), c->bres.p, sizeof(double) * 8 * (size_t)n,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms_a = 0, ms_b = 0;
    HIP_TRY(hipEventElapsedTime(&ms_a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms_b, c->ev[1], c->ev[2]));
    for (int i = 0; i < n; ++i) {
        double* h = c->h_bres.data() + 8 * (size_t)i;
        if (!american) { h[2] = 0.0; h[4] = 0.0; }
        memset(&res[i], 0, sizeof res[i]);
        fill_result(&res[i], h, p[i].n_paths);
    }
    // whole-batch times are reported on the first entry
    res[0].ms_paths = american ? ms_a : ms_b;
    res[0].ms_lsm = american ? ms_b : 0.0;
    res[0].ms_total = ms_a + ms_b;
    return 0;
}

// A batch launches ONE instantiation of every kernel: 16-byte accesses only if every problem admits them.  A
// problem's sums depend on that width (it sets the order in which a thread meets its paths), so -- for every
// problem to return the bits of its own single call whatever else is in the batch -- a mixed batch runs as two
// groups: the problems that admit 16-byte accesses, and the rest.
static bool item_vec4(const omc_params& q)
{
    const int64_t M = q.n_paths, P = (q.model == OMC_MODEL_GBM && !q.antithetic) ? M : M / 2;
    return (P % 4) == 0 && (M % 4) == 0;
}

using GroupRun = std::function<int(const omc_params*, int, omc_result*, const int*)>;
static int run_grouped(const omc_params* p, int n, omc_result* res, const GroupRun& run)
{
    int n4 = 0;
    for (int i = 0; i < n; ++i) n4 += item_vec4(p[i]) ? 1 : 0;
    if (n4 == 0 || n4 == n) return run(p, n, res, nullptr);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<omc_params> q;
        std::vector<int> idx;
        for (int i = 0; i < n; ++i)
            if (item_vec4(p[i]) == (pass == 0)) {
                q.push_back(p[i]);
                idx.push_back(i);
            }
        std::vector<omc_result> r(q.size());
        const int rc = run(q.data(), (int)q.size(), r.data(), idx.data());
        if (rc) return rc;
        for (size_t k = 0; k < idx.size(); ++k) res[idx[k]] = r[k];
    }
    // whole-batch times are reported on the first entry of a batch: add the two groups'
    return 0;
}

static int run_batch(omc_ctx* c, const omc_params* p, int n, omc_result* res, bool american)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_batch(p, n))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    return run_grouped(p, n, res, [&](const omc_params* q, int m, omc_result* r, const int*) {
        return run_batch_group(c, q, m, r, american);
    });
}

int omc_price_american_batch(omc_ctx* c, const omc_params* p, int n, omc_result* res)
{
    return run_batch(c, p, n, res, true);
}

int omc_price_european_batch(omc_ctx* c, const omc_params* p, int n, omc_result* res)
{
    return run_batch(c, p, n, res, false);
}

static int contnet_batch_group(omc_ctx* c, const omc_params* p, int n, int nn_hidden, int nn_epochs, double nn_lr,
                               const uint64_t* nn_seeds, omc_result* res)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_batch(p, n))) return rc;
    if (!res || !nn_seeds) return fail(-7, "null pointer.");
    if (p[0].semantics != OMC_SEM_REFERENCE)
        return fail(-4, "the per-step network is the regressor of the reference flow (semantics 0).");
    if ((rc = check_contnet(c, nn_hidden, nn_epochs, nn_lr))) return rc;
    const int H = omc::cn_padded_width(nn_hidden);
    const size_t slab = omc::batch_slab_bytes(p, n, true, false);
    const size_t slab2 = omc::batch_cn_slab_bytes(p, n, nn_hidden);
    const size_t nd = omc::batch_discount_doubles(p, n);
    if ((rc = c->bslab.ensure(slab))) return rc;
    if ((rc = c->cn_data.ensure(slab2))) return rc;
    if ((rc = c->btable.ensure(omc::batch_table_bytes(n)))) return rc;
    if ((rc = c->cn_scratch.ensure(omc::batch_cn_table_bytes(n)))) return rc;
    if ((rc = c->mb_table.ensure(omc::mlp_batch_table_bytes(n)))) return rc;
    if ((rc = c->cn_cont.ensure(sizeof(int) * ((size_t)n + 2)))) return rc;  // the trainer's tile prefix sums
    if ((rc = c->bres.ensure(sizeof(double) * 8 * (size_t)n))) return rc;
    if ((rc = c->bdisc.ensure(sizeof(double) * (nd ? nd : 1)))) return rc;
    // Adam's bias corrections 1 - beta^step for the `nn_epochs` steps every net takes (optim.Adam defaults)
    const double beta1 = 0.9, beta2 = 0.999;
    if (c->mb_beta1 != beta1 || c->mb_beta2 != beta2 || c->mb_bc_cap < (size_t)nn_epochs + 2) {
        const size_t cap = (size_t)nn_epochs + 1024;
        c->mb_bc_host.assign(2 * cap, 0.0);
        for (size_t k = 0; k < cap; ++k) {
            c->mb_bc_host[k] = 1.0 - std::pow(beta1, (double)k);
            c->mb_bc_host[cap + k] = 1.0 - std::pow(beta2, (double)k);
        }
        if ((rc = c->mb_bc.ensure(sizeof(double) * 2 * cap))) return rc;
        HIP_TRY(hipMemcpyAsync(c->mb_bc.p, c->mb_bc_host.data(), sizeof(double) * 2 * cap, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->mb_bc_cap = cap; c->mb_beta1 = beta1; c->mb_beta2 = beta2;
    }
    c->h_table.resize(omc::batch_table_bytes(n));
    c->h_disc.resize(nd ? nd : 1);
    c->h_bres.resize(8 * (size_t)n);
    std::vector<char> cn_table(omc::batch_cn_table_bytes(n)), mlp_table(omc::mlp_batch_table_bytes(n));
    std::vector<omc::MlpBatchJob> jobs((size_t)n);
    omc::BatchExtents e;
    omc::batch_build(p, n, true, false, (char*)c->bslab.p, (double*)c->bres.p, (double*)c->bdisc.p, c->h_table.data(),
                     c->h_disc.data(), &e);
    int max_cn_blocks = 0;
    int64_t max_paths = 0;
    omc::batch_cn_build(p, n, nn_hidden, nn_seeds, nn_lr, (char*)c->cn_data.p, c->h_table.data(), cn_table.data(),
                        jobs.data(), &max_cn_blocks, &max_paths);
    omc::mlp_batch_table_image(jobs.data(), n, H, 2, beta1, beta2, 1e-8, 0.0, 0.0, mlp_table.data());
    HIP_TRY(hipMemcpyAsync(c->btable.p, c->h_table.data(), c->h_table.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->cn_scratch.p, cn_table.data(), cn_table.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->mb_table.p, mlp_table.data(), mlp_table.size(), hipMemcpyHostToDevice, c->stream));
    if (nd)
        HIP_TRY(hipMemcpyAsync(c->bdisc.p, c->h_disc.data(), sizeof(double) * nd, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->cn_data.p, 0, sizeof(double) * 8 * (size_t)n, c->stream));  // the problems' headers
    HIP_TRY(hipStreamSynchronize(c->stream));  // cn_table / mlp_table are local pageable vectors
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(omc::batch_paths(c->stream, c->btable.p, n, e, generator_id(&p[0])));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(omc::batch_contnet(c->stream, c->btable.p, c->cn_scratch.p, c->mb_table.p, n, e, nn_hidden, nn_epochs,
                               max_cn_blocks, max_paths, (const double*)c->mb_bc.p,
                               (const double*)c->mb_bc.p + c->mb_bc_cap, (int*)c->cn_cont.p));
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_bres.data(), c->bres.p, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms_a = 0, ms_b = 0;
    HIP_TRY(hipEventElapsedTime(&ms_a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms_b, c->ev[1], c->ev[2]));
    for (int i = 0; i < n; ++i) {
        memset(&res[i], 0, sizeof res[i]);
        fill_result(&res[i], c->h_bres.data() + 8 * (size_t)i, p[i].n_paths);
    }
    res[0].ms_paths = ms_a;  // whole-batch times on the first entry
    res[0].ms_lsm = ms_b;
    res[0].ms_total = ms_a + ms_b;
    return 0;
}

int omc_price_american_contnet_batch(omc_ctx* c, const omc_params* p, int n, int nn_hidden, int nn_epochs, double nn_lr,
                                     const uint64_t* nn_seeds, omc_result* res)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_batch(p, n))) return rc;
    if (!res || !nn_seeds) return fail(-7, "null pointer.");
    return run_grouped(p, n, res, [&](const omc_params* q, int m, omc_result* r, const int* idx) {
        std::vector<uint64_t> sd((size_t)m);
        for (int k = 0; k < m; ++k) sd[(size_t)k] = nn_seeds[idx ? idx[k] : k];
        return contnet_batch_group(c, q, m, nn_hidden, nn_epochs, nn_lr, sd.data(), r);
    });
}

int omc_mlp_param_count(int hidden, int layers) { return omc::mlp_apply_param_count(hidden, layers); }

int omc_mlp_train_supported(int hidden, int layers, int64_t batch)
{
    return omc::mlp_train_kernel_choice(hidden, layers, batch) != 0;
}

int omc_lsm_apply_mlp(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                      double r, double T, int is_put, int hidden, int layers, const float* params,
                      const double* feat_mean, const double* feat_std, double y_mean, double y_std,
                      double dropout, uint64_t seed, omc_result* res, float* sx_out, int32_t* tex_out)
{
    return omc_lsm_apply_mlp_shard(c, S, ld, n_paths, n_steps, K, r, T, is_put, hidden, layers, params, feat_mean, feat_std,
                                   y_mean, y_std, dropout, seed, res, sx_out, tex_out, 0, n_paths / 2);
}

int omc_lsm_apply_mlp_shard(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                            double r, double T, int is_put, int hidden, int layers, const float* params,
                            const double* feat_mean, const double* feat_std, double y_mean, double y_std,
                            double dropout, uint64_t seed, omc_result* res, float* sx_out, int32_t* tex_out,
                            int64_t col_base0, int64_t col_base1)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (omc_mlp_param_count(hidden, layers) < 0)
        return fail(-9, "pass 2 supports hidden = 32, 64 or 128 with 2 or 3 hidden layers.");
    if (!params || !feat_mean || !feat_std || !res) return fail(-7, "null pointer.");
    if (!(dropout >= 0.0 && dropout < 1.0)) return fail(-4, "dropout must be in [0, 1).");
    for (int i = 0; i < 7; ++i)
        if (!(feat_std[i] > 0.0)) return fail(-4, "feature standard deviations must be positive.");
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, n_paths, n_steps, r, T, false, true, &w))) return rc;
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    HIP_TRY(omc::mlp_apply_pass2(c->stream, p, hidden, layers, params, feat_mean, feat_std, y_mean, y_std,
                                 dropout, seed, w.sx, w.tex, col_base0, col_base1));
    HIP_TRY(omc::lsm_final_reduce(c->stream, p, w, 1));
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = copy_outputs(c, w, n_paths, n_steps, nullptr, sx_out, tex_out))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(res, 0, sizeof *res);
    fill_result(res, c->hres, n_paths);
    return 0;
}

int omc_localvol_param_count(int hidden, int layers) { return omc::localvol_param_count(hidden, layers); }

int omc_localvol_paths_f32(omc_ctx* c, float* S, int64_t ld, int64_t n_paths, int n_steps, double S0,
                           double r, double T, double K, int hidden, int layers, const float* params,
                           double m_scale, double tau_scale, double epsilon, const float* Z)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(S0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (omc_localvol_param_count(hidden, layers) < 0)
        return fail(-9, "the local-vol kernel supports hidden_dim = 64 with 1..8 hidden layers.");
    if (!params || !Z) return fail(-7, "null pointer.");
    if (n_paths & 1) return fail(-3, "antithetic layout needs an even n_paths.");
    if (!(m_scale > 0) || !(tau_scale > 0)) return fail(-4, "scaler values must be positive.");
    HIP_TRY(omc::localvol_paths(c->stream, S, ld, n_paths, n_steps, layers, params, Z, S0, r, T, K, m_scale,
                                tau_scale, epsilon));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_nn_build_rows(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, double r,
                      double T, int is_put, float* data, int64_t cap_rows, int64_t* n_rows, double* stats16)
{
    int rc;
    const RowsCache had = c ? c->rows_cache : RowsCache{};
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (!n_rows || (data && !stats16)) return fail(-7, "null pointer.");
    const bool full = data != nullptr;
    // The count call (data == NULL) already makes the ONE sweep that counts and forms the statistics; when the call with
    // `data` is the very next call on this context with the same arguments, it starts from those results (counts and
    // offsets are still in the context's scratch) and only writes the rows: S is read twice in all, not three times.
    const bool hit = full && had.valid && had.S == S && had.ld == ld && had.M == n_paths && had.N == n_steps &&
                     had.is_put == (is_put ? 1 : 0) && had.K == K && had.r == r && had.T == T;
    // On a context with a communicator / hook the full call is COLLECTIVE (two small all-reduces below).  A failure that
    // only this rank can see -- no memory for its scratch, a row buffer too small for ITS rows, a HIP error -- must not
    // send it home before the peers have entered them: it is carried as a flag in the first all-reduce instead, and
    // every rank of the job returns an error together.
    const bool dist = full && c->distributed();
    int lerr = 0;
    std::string ltext;
    auto local_failure = [&](int code) {
        lerr = code;
        ltext = g_err;
    };
    omc::LsmWorkspace w;
    if ((rc = prepare_lsm(c, n_paths, n_steps, r, T, false, false, &w)) ||
        (rc = c->scratch.ensure(omc::nn_rows_scratch_bytes(n_paths, n_steps)))) {
        if (!dist) return rc;
        local_failure(rc);
    }
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    int64_t R = 0;
    double st[16] = {0.0};  // n, mean[7], M2[7] of [x, x^2, x^3, max(x-1,0), s, x*s, y] over this rank's rows
    if (!lerr && hit) {
        R = had.R;
        memcpy(st, had.st, sizeof st);
    } else if (!lerr) {
        // ONE sweep over S: the counts of every (step, tile) and the statistics
        const int64_t* total_dev = nullptr;
        const double* stats_dev = nullptr;
        hipError_t e = omc::nn_rows_count(c->stream, p, w.D, c->scratch.p, &total_dev, &stats_dev);
        if (e == hipSuccess) e = hipMemcpyAsync(&R, total_dev, sizeof R, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(st, stats_dev, sizeof st, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            g_err = std::string("pass 1 of the NN flow failed: ") + hipGetErrorString(e);
            if (!dist) return (int)e;
            local_failure((int)e);
            R = 0;
        }
    }
    *n_rows = R;
    if (!full) {  // count only: leave the sweep's results for the call with `data`
        RowsCache& k = c->rows_cache;
        k.valid = true; k.S = S; k.ld = ld; k.M = n_paths; k.N = n_steps; k.is_put = is_put ? 1 : 0; k.K = K; k.r = r; k.T = T;
        k.R = R;
        memcpy(k.st, st, sizeof st);
        return 0;
    }
    if (!lerr && cap_rows < R) {
        g_err = "row buffer smaller than the number of in-the-money (step, path) pairs.";
        if (!dist) return -6;
        local_failure(-6);
    }
    for (int i = 0; i < 16; ++i) stats16[i] = i < 7 ? 0.0 : 1.0;
    stats16[0] = 1.0;  // the constant feature: mean 1, std 0 -> 1
    stats16[14] = 0.0;
    double mean[7], m2[7], Rg = (double)R;
    if (!dist) {
        if (R == 0) return 0;
        for (int q = 0; q < 7; ++q) {
            mean[q] = st[1 + q];
            m2[q] = st[8 + q];
        }
    } else {
        // The statistics are those of ALL ranks' rows (the reference trains one network on the rows of all paths):
        // (1) sum over the ranks of n_r mean_r and n_r -> the global means; (2) sum of M2_r + n_r (mean_r - mean)^2 ->
        // the global sum of squared deviations (Chan's merge, for any number of ranks at once).  A rank without rows
        // contributes zeros; when NO rank has a row every rank returns the defaults together.
        if ((rc = c->seq_vote.ensure(sizeof(double) * 16))) return rc;  // (never allocates: exists since the communicator / hook was installed)
        const double nr = lerr ? 0.0 : st[0];
        double v[9];
        for (int q = 0; q < 7; ++q) v[q] = nr * st[1 + q];
        v[7] = nr;
        v[8] = lerr ? 1.0 : 0.0;
        if ((rc = allreduce_host(c, (double*)c->seq_vote.p, v, 9))) return rc;
        if (v[8] > 0.0) {
            if (lerr) return fail(lerr, ltext.c_str());
            return fail(3102, "another rank of the job could not build its training rows.");
        }
        Rg = v[7];
        if (!(Rg > 0.0)) return 0;
        double dv[8];
        for (int q = 0; q < 7; ++q) {
            mean[q] = v[q] / Rg;
            const double dm = st[1 + q] - mean[q];
            dv[q] = nr > 0.0 ? st[8 + q] + nr * dm * dm : 0.0;
        }
        dv[7] = 0.0;
        if ((rc = allreduce_host(c, (double*)c->seq_vote.p, dv, 8))) return rc;
        for (int q = 0; q < 7; ++q) m2[q] = dv[q];
    }
    // layout: feat_mean[0..6], feat_std[7..13], y_mean [14], y_std [15]; population std, zero std -> 1 (:551-563)
    for (int q = 0; q < 6; ++q) {
        const double sd = std::sqrt(m2[q] / Rg);
        stats16[1 + q] = mean[q];
        stats16[8 + q] = sd > 1e-13 * std::fabs(mean[q]) ? sd : 1.0;
    }
    const double ysd = std::sqrt(m2[6] / Rg);
    stats16[14] = mean[6];
    stats16[15] = ysd > 1e-13 * std::fabs(mean[6]) ? ysd : 1.0;
    if (R > 0) {
        HIP_TRY(omc::nn_rows_write(c->stream, p, w.D, c->scratch.p, stats16, stats16 + 7, stats16[14], stats16[15], data,
                                   cap_rows));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// Solve the normal equations of the live (non-constant) standardised columns: A w = b with A = the correlation matrix of
// the columns, b = their correlations with the target; symmetric Gauss elimination with diagonal pivoting, a column
// whose pivot has vanished (exactly collinear with those before it) gets weight 0 -- any solution of a consistent
// singular system predicts the same values, and that is all pass 2 uses.
static void ols7_solve(const double* st, double n, const double* sd, const bool* live, double* w7)
{
    auto C = [&](int i, int j) { return i <= j ? st[8 + i * 7 - i * (i - 1) / 2 + (j - i)] : st[8 + j * 7 - j * (j - 1) / 2 + (i - j)]; };
    int idx[6], m = 0;
    for (int q = 0; q < 6; ++q)
        if (live[q]) idx[m++] = q;
    double A[6][7];
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < m; ++j) A[i][j] = C(idx[i], idx[j]) / (n * sd[idx[i]] * sd[idx[j]]);
        A[i][m] = C(idx[i], 6) / (n * sd[idx[i]] * sd[6]);
    }
    bool used[6] = {false, false, false, false, false, false};
    int order[6], rank = 0;
    for (int k = 0; k < m; ++k) {
        int pv = -1;
        double best = 1e-13;  // (unit diagonal: a pivot below this is rounding noise of an exactly dependent column)
        for (int i = 0; i < m; ++i)
            if (!used[i] && A[i][i] > best) { best = A[i][i]; pv = i; }
        if (pv < 0) break;
        used[pv] = true;
        order[rank++] = pv;
        for (int i = 0; i < m; ++i) {
            if (i == pv) continue;
            const double f = A[i][pv] / A[pv][pv];
            if (f == 0.0) continue;
            for (int j = 0; j <= m; ++j) A[i][j] -= f * A[pv][j];
        }
    }
    for (int q = 0; q < 7; ++q) w7[q] = 0.0;  // column 0 is the constant: normalised to zero, minimum-norm weight 0
    for (int k = 0; k < rank; ++k) w7[1 + idx[order[k]]] = A[order[k]][m] / A[order[k]][order[k]];
}

// The body of omc_lsm_ols7 / omc_price_american_ols7 behind their argument checks (which are the same on every rank).
// On a context with a communicator / hook the call is COLLECTIVE (two small all-reduces for the fit, one for the result).
// A failure only this rank can see -- no memory for its path matrix or workspace (`pre_err` / `pre_text`: what the caller
// already ran into), a HIP error in its sweep or in its pass 2 -- travels as a flag: in the ninth double of the first
// all-reduce, resp. in slot 7 of the result sums (which the kernels leave at zero), and every rank returns an error
// together (the rank's own code there, 3103 on its peers) instead of leaving the peers inside a collective.
static int lsm_ols7_run(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, double r, double T,
                        int is_put, omc_result* res, double* weights7, double* stats16, float* sx_out, int32_t* tex_out,
                        int pre_err, const std::string& pre_text)
{
    int rc;
    const bool dist = c->distributed();
    int lerr = pre_err;
    std::string ltext = pre_text;
    auto local_failure = [&](int code) {
        lerr = code;
        ltext = g_err;
    };
    omc::LsmWorkspace w;
    if (!lerr && ((rc = prepare_lsm(c, n_paths, n_steps, r, T, false, true, &w)) ||
                  (rc = c->scratch.ensure(omc::ols7_scratch_bytes(n_paths, n_steps))))) {
        if (!dist) return rc;
        local_failure(rc);
    }
    if (lerr && !dist) return fail(lerr, ltext.c_str());
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, r, T};
    // pass 1 (:482-516): one sweep -> (n, mean, co-moments) of the 6 non-constant features and the target
    double st[omc::kOls7Stats] = {0.0};
    if (!lerr) {
        const double* stats_dev = nullptr;
        hipError_t e = omc::ols7_comoments(c->stream, p, w.D, c->scratch.p, &stats_dev);
        if (e == hipSuccess) e = hipMemcpyAsync(st, stats_dev, sizeof st, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            g_err = std::string("the co-moment sweep failed: ") + hipGetErrorString(e);
            if (!dist) return (int)e;
            local_failure((int)e);
        }
    }
    if (dist) {
        // The fit is over ALL ranks' rows (paths shard by antithetic pair, the regression does not): the ranks' triples
        // (n_r, mean_r, C_r) are merged by Chan's formula for any number of ranks at once -- (1) sum of n_r mean_r and
        // n_r -> the global means; (2) sum of C_r + n_r (mean_r - mean)(mean_r - mean)^T -> the global co-moments -- the
        // "regression moments" all-reduce of north_star, 9 + 28 doubles.  Every rank then solves the same 6 x 6 system.
        if ((rc = c->seq_vote.ensure(sizeof(double) * 40))) return rc;  // (never allocates: exists since the communicator / hook was installed)
        const double nr = lerr ? 0.0 : st[0];
        double v[9], mr[7];
        for (int q = 0; q < 7; ++q) {
            mr[q] = st[1 + q];
            v[q] = nr * mr[q];
        }
        v[7] = nr;
        v[8] = lerr ? 1.0 : 0.0;
        if ((rc = allreduce_host(c, (double*)c->seq_vote.p, v, 9))) return rc;
        if (v[8] > 0.0) {
            if (lerr) return fail(lerr, ltext.c_str());
            return fail(3103, "another rank of the job could not run its co-moment sweep.");
        }
        const double ng = v[7];
        double cg[28];
        for (int i = 0, k = 0; i < 7; ++i)
            for (int j = i; j < 7; ++j, ++k) {
                const double di = ng > 0.0 ? mr[i] - v[i] / ng : 0.0, dj = ng > 0.0 ? mr[j] - v[j] / ng : 0.0;
                cg[k] = nr > 0.0 ? st[8 + k] + nr * di * dj : 0.0;
            }
        if ((rc = allreduce_host(c, (double*)c->seq_vote.p, cg, 28))) return rc;
        st[0] = ng;
        for (int q = 0; q < 7; ++q) st[1 + q] = ng > 0.0 ? v[q] / ng : 0.0;
        for (int k = 0; k < 28; ++k) st[8 + k] = cg[k];
    }
    const double n = st[0];
    // normalisation (:550-563): population std, zero std -> 1 (the column is then all zero)
    double s16[16], sd[7], w7[7];
    bool live[7];
    for (int i = 0; i < 16; ++i) s16[i] = i < 7 ? 0.0 : 1.0;
    s16[0] = 1.0;
    s16[14] = 0.0;
    for (int q = 0; q < 7; ++q) w7[q] = 0.0;
    if (n > 0.0) {
        // The sweep's quantities are g = [u, u^2, u^3, max(u, 0), s, u s, y] with u = x - 1 (what the per-step polynomial uses):
        // the same span as the reference's features f = [x, x^2, x^3, max(x - 1, 0), s, x s] plus the constant -- the same
        // fit -- but a far better conditioned Gram matrix for in-the-money spots, which sit within a few tens of percent
        // of the strike (the normal equations carry eps * cond^2).  f = B g + b with a unit lower-triangular B:
        //   x = u + 1, x^2 = u^2 + 2 u + 1, x^3 = u^3 + 3 u^2 + 3 u + 1, x s = u s + s.
        // The system is solved for g; means, stds and weights are then stated for f, the reference's features.
        static const double B[6][6] = {{1, 0, 0, 0, 0, 0}, {2, 1, 0, 0, 0, 0}, {3, 3, 1, 0, 0, 0},
                                       {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 1, 1}};
        static const double b0[6] = {1, 1, 1, 0, 0, 0};
        auto Cg = [&](int i, int j) { return i <= j ? st[8 + i * 7 - i * (i - 1) / 2 + (j - i)] : st[8 + j * 7 - j * (j - 1) / 2 + (i - j)]; };
        double sdg[7];
        bool liveg[7];
        for (int q = 0; q < 7; ++q) {
            const double mean = st[1 + q], v = std::sqrt(Cg(q, q) / n);
            // (u is centred near 0: its own size is no yardstick for "constant" -- the spot's is, x = u + 1)
            const double scale = q < 3 ? 1.0 : std::fabs(mean);
            liveg[q] = v > 1e-13 * scale;
            sdg[q] = liveg[q] ? v : 1.0;
        }
        double wg[7];
        for (int q = 0; q < 7; ++q) wg[q] = 0.0;
        if (liveg[6]) ols7_solve(st, n, sdg, liveg, wg);  // (a constant target: every weight 0, continuation = its mean)
        // the reference's features: means, population stds (zero -> 1, :562), and the weights c = B^-T a, a = wg / sdg
        double a[6], cf[6];
        for (int q = 0; q < 6; ++q) a[q] = wg[1 + q] / sdg[q];
        cf[5] = a[5];
        cf[4] = a[4] - cf[5];
        cf[3] = a[3];
        cf[2] = a[2];
        cf[1] = a[1] - 3.0 * cf[2];
        cf[0] = a[0] - 2.0 * cf[1] - 3.0 * cf[2];
        for (int i = 0; i < 6; ++i) {
            double mean = b0[i], var = 0.0;
            for (int j = 0; j < 6; ++j) {
                mean += B[i][j] * st[1 + j];
                for (int k = 0; k < 6; ++k) var += B[i][j] * B[i][k] * Cg(j, k);
            }
            const double v = std::sqrt(std::fmax(var, 0.0) / n);
            live[i] = v > 1e-13 * std::fabs(mean);
            sd[i] = live[i] ? v : 1.0;
            s16[1 + i] = mean;
            s16[8 + i] = sd[i];
            w7[1 + i] = live[i] ? cf[i] * sd[i] : 0.0;  // (a constant column contributes (f - mean) = 0 whatever its weight)
        }
        live[6] = liveg[6];
        sd[6] = sdg[6];
        s16[14] = st[7];
        s16[15] = sd[6];
    }
    // pass 2 (:615-651) with the fit, then the mean of the cash-flows valued at t = dt (:651)
    hipError_t e2 = omc::ols7_pass2(c->stream, p, s16, s16 + 7, w7, s16[14], s16[15], w.sx, w.tex);
    if (e2 == hipSuccess) e2 = omc::lsm_final_reduce(c->stream, p, w, 1);
    if (e2 != hipSuccess) {
        g_err = std::string("pass 2 with the fit failed: ") + hipGetErrorString(e2);
        if (!dist) return (int)e2;
        local_failure((int)e2);
        static const double kOne = 1.0;  // this rank's flag rides in slot 7 of the sums about to be all-reduced
        (void)hipMemcpyAsync(w.result + 7, &kOne, sizeof kOne, hipMemcpyHostToDevice, c->stream);
    }
    if (dist && (rc = allreduce(c, w.result, 8))) return rc;  // the discounted-payoff sums of all ranks (+ the flag)
    HIP_TRY(hipMemcpyAsync(c->hres, w.result, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if (!lerr && (rc = copy_outputs(c, w, n_paths, n_steps, nullptr, sx_out, tex_out))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (dist && (lerr || c->hres[7] > 0.0)) {
        if (lerr) return fail(lerr, ltext.c_str());
        return fail(3103, "another rank of the job could not run its pass 2.");
    }
    memset(res, 0, sizeof *res);
    fill_result(res, c->hres, c->distributed() ? n_paths * c->world : n_paths, c->distributed() ? c->world : 1);
    res->sum_nitm = (int64_t)llround(n);  // rows of the regression (of the job)
    if (weights7) memcpy(weights7, w7, sizeof w7);
    if (stats16) memcpy(stats16, s16, sizeof s16);
    return 0;
}

int omc_lsm_ols7(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, double r, double T,
                 int is_put, omc_result* res, double* weights7, double* stats16, float* sx_out, int32_t* tex_out)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_market(1.0, K, T, r))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    return lsm_ols7_run(c, S, ld, n_paths, n_steps, K, r, T, is_put, res, weights7, stats16, sx_out, tex_out, 0, std::string());
}

int omc_price_american_ols7(omc_ctx* c, const omc_params* p, omc_result* res, double* weights7, double* stats16)
{
    int rc;
    if ((rc = bind(c))) return rc;
    if ((rc = check_params(p))) return rc;
    if (!res) return fail(-7, "null result pointer.");
    const int64_t ld = (p->n_paths + 63) / 64 * 64;
    // the context's own path matrix: the largest allocation of the call and the likeliest to fail on a card shared with
    // other tenants -- on a distributed context that failure must reach the peers (lsm_ols7_run), not strand them
    int pre = 0;
    std::string text;
    if ((rc = c->S.ensure(sizeof(float) * (size_t)ld * (size_t)(p->n_steps + 1))) ||
        (rc = enqueue_paths(c, p, (float*)c->S.p, ld))) {
        if (!c->distributed()) return rc;
        pre = rc;
        text = g_err;
    }
    return lsm_ols7_run(c, (const float*)c->S.p, ld, p->n_paths, p->n_steps, p->K, p->r, p->T, p->is_put ? 1 : 0, res, weights7,
                        stats16, nullptr, nullptr, pre, text);
}

int omc_nn_feature_stats(omc_ctx* c, const double* x, const int32_t* t, const double* y, int64_t n_rows,
                         double T, double dt, double* out16)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (!x || !t || !y || !out16) return fail(-7, "null pointer.");
    if (n_rows <= 0) return fail(-3, "n_rows must be positive.");
    if ((rc = c->scratch.ensure(omc::nn_stats_scratch_bytes() + sizeof(double) * 16))) return rc;
    double* scratch = (double*)c->scratch.p;
    double* dev16 = scratch + omc::nn_stats_scratch_bytes() / sizeof(double);
    HIP_TRY(omc::nn_feature_stats(c->stream, x, t, y, n_rows, T, dt, scratch, dev16));
    HIP_TRY(hipMemcpyAsync(out16, dev16, sizeof(double) * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_mlp_train_variant(int hidden, int layers, int64_t batch)
{
    return omc::mlp_train_kernel_choice(hidden, layers, batch);
}

int omc_mlp_dropout_masks(omc_ctx* c, int variant, int hidden, int layers, int64_t n_rows, const uint32_t* keys,
                          uint32_t step, uint64_t seed, double dropout, uint8_t* out)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (variant < 0 || variant > 4) return fail(-4, "variant must be 0 (pass 2) or 1 .. 4 (omc_mlp_train_variant).");
    const bool shape_ok = (variant == 3 || variant == 0) ? (hidden == 32 || hidden == 64 || hidden == 128)
                        : variant == 1 ? hidden == 64 : (hidden == 64 || hidden == 128);
    if (!shape_ok || layers < 1 || layers > 3) return fail(-9, "this kernel does not exist for that network shape.");
    if (n_rows <= 0 || !out) return fail(-3, "n_rows must be positive, out non-null.");
    if (!(dropout >= 0.0 && dropout < 1.0)) return fail(-4, "dropout must be in [0, 1).");
    const size_t nout = (size_t)layers * (size_t)n_rows * (size_t)hidden, nkey = keys ? sizeof(uint32_t) * (size_t)n_rows : 0;
    if ((rc = c->scratch.ensure(nout + nkey + 16))) return rc;
    uint8_t* dout = (uint8_t*)c->scratch.p;
    uint32_t* dkeys = keys ? (uint32_t*)(dout + ((nout + 15) & ~(size_t)15)) : nullptr;
    if (keys) HIP_TRY(hipMemcpyAsync(dkeys, keys, nkey, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(omc::mlp_dropout_masks(c->stream, variant, hidden, layers, n_rows, dkeys, step, seed, dropout, dout));
    HIP_TRY(hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_mlp_shuffle_indices(omc_ctx* c, int64_t n_rows, uint64_t shuffle_key, int64_t* out_device)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (n_rows <= 0 || !out_device) return fail(-3, "n_rows must be positive, out non-null.");
    HIP_TRY(omc::mlp_shuffle_indices(c->stream, n_rows, shuffle_key, out_device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_mlp_train_epoch(omc_ctx* c, const float* data, int64_t n_rows, int64_t batch, int hidden,
                        int layers, float* params, float* adam_m, float* adam_v, int64_t* step,
                        double lr, double beta1, double beta2, double eps, double weight_decay,
                        double dropout, uint64_t seed, uint64_t shuffle_key, double* mean_loss)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (omc::mlp_train_kernel_choice(hidden, layers, batch) == 0)
        return fail(-9, "the fused trainer supports hidden = 32, 64 or 128 with 2 or 3 hidden layers.");
    if (!data || !params || !adam_m || !adam_v || !step || !mean_loss) return fail(-7, "null pointer.");
    if (n_rows <= 0 || batch <= 0 || *step < 0) return fail(-3, "n_rows, batch must be positive.");
    if (!(dropout >= 0.0 && dropout < 1.0)) return fail(-4, "dropout must be in [0, 1).");
    if (!(lr > 0.0)) return fail(-4, "learning rate must be positive.");
    if ((rc = c->mlp_part.ensure(omc::mlp_partial_bytes(hidden, layers, batch)))) return rc;
    if ((rc = c->mlp_wt.ensure(omc::mlp_wt_bytes(hidden, layers)))) return rc;
    if ((rc = c->mlp_loss.ensure(sizeof(double)))) return rc;
    HIP_TRY(hipMemsetAsync(c->mlp_loss.p, 0, sizeof(double), c->stream));
    omc::MlpTrainPlan t;
    t.data = data; t.params = params; t.adam_m = adam_m; t.adam_v = adam_v;
    t.partial = (float*)c->mlp_part.p; t.loss_acc = (double*)c->mlp_loss.p;
    t.nrows = n_rows; t.batch = batch; t.first_step = *step; t.hidden = hidden; t.layers = layers;
    t.wt = (float*)c->mlp_wt.p;
    t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.weight_decay = weight_decay;
    t.dropout = dropout; t.seed = seed; t.shuffle_key = shuffle_key;
    t.allow_q16 = true;
    const int64_t nb = (n_rows + batch - 1) / batch;
    HIP_TRY(omc::mlp_train_steps(c->stream, t));
    double acc = 0.0;
    HIP_TRY(hipMemcpyAsync(&acc, c->mlp_loss.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *step += nb;
    *mean_loss = acc / (double)nb;
    return 0;
}

// ---- the NN regressor sharded over the ranks of a job (options_model_3.py:542-613)
int omc_nn_half_counts(omc_ctx* c, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, int is_put,
                       int64_t* counts)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if ((rc = check_sizes(n_paths, n_steps))) return rc;
    if ((rc = check_matrix(S, ld, n_paths))) return rc;
    if (!counts) return fail(-7, "null pointer.");
    if (n_steps < 2) return 0;
    const size_t bytes = sizeof(int64_t) * 2 * (size_t)(n_steps - 1);
    if ((rc = c->scratch.ensure(bytes))) return rc;
    omc::LsmProblem p{S, ld, n_paths, n_steps, is_put ? 1 : 0, K, 0.0, 1.0};
    HIP_TRY(omc::nn_rows_half_counts(c->stream, p, n_paths / 2, (int64_t*)c->scratch.p));
    HIP_TRY(hipMemcpyAsync(counts, c->scratch.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int omc_mlp_shard_epoch(omc_ctx* c, const float* data, int64_t n_rows_local, int64_t rows_global, int64_t batch,
                        uint64_t shuffle_key, const int64_t* gstart, const int64_t* lstart, int nseg, int segs_per_step,
                        float* data_epoch, uint32_t* drop_pos, int64_t* step_off)
{
    int rc;
    if ((rc = bind_in(c))) return rc;
    if (!gstart || !lstart || !step_off || (n_rows_local > 0 && (!data || !data_epoch || !drop_pos)))
        return fail(-7, "null pointer.");
    if (n_rows_local < 0 || rows_global <= 0 || batch <= 0 || nseg <= 0 || n_rows_local > rows_global)
        return fail(-3, "row counts, batch and segment count must be positive.");
    if (gstart[0] != 0 || gstart[nseg] != rows_global) return fail(-4, "segment table does not cover [0, rows_global).");
    const int64_t steps = (rows_global + batch - 1) / batch;
    // scratch: segment tables | selection scan | sel_row, sel_i | step offsets
    const size_t tab = sizeof(int64_t) * (size_t)(2 * nseg + 1), scan = omc::mlp_shard_scratch_bytes(rows_global),
                 sel = sizeof(int64_t) * (size_t)(n_rows_local + 1), so = sizeof(int64_t) * (size_t)(steps + 1);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    if ((rc = c->shard.ensure(up(tab) + up(scan) + 2 * up(sel) + up(so)))) return rc;
    char* b = (char*)c->shard.p;
    int64_t* d_g = (int64_t*)b;
    int64_t* d_l = d_g + nseg + 1;
    void* d_scan = b + up(tab);
    int64_t* sel_row = (int64_t*)(b + up(tab) + up(scan));
    int64_t* sel_i = (int64_t*)((char*)sel_row + up(sel));
    int64_t* d_so = (int64_t*)((char*)sel_i + up(sel));
    HIP_TRY(hipMemcpyAsync(d_g, gstart, sizeof(int64_t) * (size_t)(nseg + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_l, lstart, sizeof(int64_t) * (size_t)nseg, hipMemcpyHostToDevice, c->stream));
    // own segments must tile [0, n_rows_local) of the rank's matrix: checked here, on the host, before any kernel
    // indexes `data` with them
    {
        std::vector<std::pair<int64_t, int64_t>> own;
        for (int s = 0; s < nseg; ++s) {
            if (gstart[s + 1] < gstart[s]) return fail(-4, "segment table is not ascending.");
            if (lstart[s] >= 0 && gstart[s + 1] > gstart[s]) own.push_back({lstart[s], gstart[s + 1] - gstart[s]});
        }
        std::sort(own.begin(), own.end());
        int64_t at = 0;
        for (auto& o : own) {
            if (o.first != at) return fail(-4, "this rank's segments do not tile its rows.");
            at += o.second;
        }
        if (at != n_rows_local) return fail(-4, "this rank's segments do not add up to its row count.");
    }
    const int64_t* total_dev = nullptr;
    const int group = (segs_per_step > 0 && nseg % segs_per_step == 0) ? segs_per_step : 0;
    HIP_TRY(omc::mlp_shard_select(c->stream, rows_global, shuffle_key, d_g, d_l, nseg, group, d_scan, sel_row, sel_i, &total_dev));
    HIP_TRY(omc::mlp_shard_gather(c->stream, data, sel_row, sel_i, n_rows_local, batch, steps, data_epoch, drop_pos, d_so));
    int64_t total = -1;
    HIP_TRY(hipMemcpyAsync(&total, total_dev, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(step_off, d_so, sizeof(int64_t) * (size_t)(steps + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (total != n_rows_local) return fail(-4, "the epoch's permutation selected another number of rows than the rank owns.");
    return 0;
}

static int allreduce_cb(void* user, double* dptr, int count) { return allreduce((omc_ctx*)user, dptr, count); }

int omc_mlp_train_epoch_sharded(omc_ctx* c, const float* data_epoch, int64_t n_rows_local, int64_t rows_global,
                                int64_t batch, int hidden, int layers, float* params, float* adam_m, float* adam_v,
                                int64_t* step, double lr, double beta1, double beta2, double eps, double weight_decay,
                                double dropout, uint64_t seed, const int64_t* step_off, const uint32_t* drop_pos,
                                double* mean_loss)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (!params || !adam_m || !adam_v || !step || !mean_loss || !step_off || (n_rows_local > 0 && (!data_epoch || !drop_pos)))
        return fail(-7, "null pointer.");
    if (n_rows_local < 0 || rows_global <= 0 || batch <= 0 || *step < 0) return fail(-3, "row counts, batch must be positive.");
    if (!(dropout >= 0.0 && dropout < 1.0)) return fail(-4, "dropout must be in [0, 1).");
    if (!(lr > 0.0)) return fail(-4, "learning rate must be positive.");
    const int64_t steps = (rows_global + batch - 1) / batch;
    if (step_off[0] != 0 || step_off[steps] != n_rows_local) return fail(-4, "step offsets do not cover the rank's rows.");
    for (int64_t k = 0; k < steps; ++k)
        if (step_off[k + 1] < step_off[k] || step_off[k + 1] - step_off[k] > batch) return fail(-4, "step offsets are not ascending.");
    omc::MlpTrainPlan t;
    t.data = data_epoch; t.params = params; t.adam_m = adam_m; t.adam_v = adam_v;
    t.nrows = n_rows_local; t.batch = batch; t.first_step = *step; t.hidden = hidden; t.layers = layers;
    t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.weight_decay = weight_decay;
    t.dropout = dropout; t.seed = seed; t.shuffle_key = 0;
    t.allow_q16 = true;
    t.step_off = step_off; t.rows_global = rows_global; t.drop_pos = drop_pos;
    t.allreduce = allreduce_cb; t.allreduce_user = c;  // no communicator / hook: the sum of one rank
    const int64_t kb = omc::mlp_plan_kernel_batch(t);
    if (omc::mlp_train_kernel_choice(hidden, layers, kb) == 0)
        return fail(-9, "the fused trainer supports hidden = 32, 64 or 128 with 2 or 3 hidden layers.");
    const int np = omc::mlp_train_param_count(hidden, layers);
    if ((rc = c->mlp_part.ensure(omc::mlp_partial_bytes(hidden, layers, kb)))) return rc;
    if ((rc = c->mlp_wt.ensure(omc::mlp_wt_bytes(hidden, layers)))) return rc;
    if ((rc = c->mlp_loss.ensure(sizeof(double)))) return rc;
    if ((rc = c->mlp_gred.ensure((sizeof(double) + sizeof(float)) * (size_t)(np + 1)))) return rc;
    HIP_TRY(hipMemsetAsync(c->mlp_loss.p, 0, sizeof(double), c->stream));
    t.partial = (float*)c->mlp_part.p; t.loss_acc = (double*)c->mlp_loss.p; t.wt = (float*)c->mlp_wt.p;
    t.gred = (double*)c->mlp_gred.p;
    HIP_TRY(omc::mlp_train_steps(c->stream, t));
    double acc = 0.0;
    HIP_TRY(hipMemcpyAsync(&acc, c->mlp_loss.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *step += steps;
    *mean_loss = acc / (double)steps;
    return 0;
}

int omc_mlp_train_batch_supported(int hidden, int layers, int64_t batch)
{
    return omc::mlp_batch_supported(hidden, layers, batch) ? 1 : 0;
}

int omc_mlp_train_epoch_batch(omc_ctx* c, omc_mlp_job* jobs, int n, int hidden, int layers, double beta1, double beta2,
                              double eps, double weight_decay, double dropout)
{
    int rc = bind_in(c);
    if (rc) return rc;
    if (!jobs || n <= 0) return fail(-7, "empty batch.");
    if (n > 65535) return fail(-3, "batch too large (max 65535 networks per call).");
    if (!(dropout >= 0.0 && dropout < 1.0)) return fail(-4, "dropout must be in [0, 1).");
    int64_t max_steps = 0, max_batch32 = 0, max_batch16 = 0, last_step = 0;
    size_t part_bytes = 0;
    const int64_t q16_rows = omc::mlp_q16_rows(hidden);
    for (int i = 0; i < n; ++i) {
        const omc_mlp_job& j = jobs[i];
        if (!j.data || !j.params || !j.adam_m || !j.adam_v) return fail(-7, "null pointer.");
        if (j.n_rows <= 0 || j.batch <= 0 || j.step < 0) return fail(-3, "n_rows, batch must be positive.");
        if (!(j.lr > 0.0)) return fail(-4, "learning rate must be positive.");
        if (!omc::mlp_batch_supported(hidden, layers, j.batch))
            return fail(-9, "the batched trainer covers the one-tile-per-workgroup shapes (64 | 128 units x 2 | 3 layers at "
                            "minibatches of at most 8192 rows, 32 units x 2 | 3 layers).");
        const int64_t nb = (j.n_rows + j.batch - 1) / j.batch;
        if (nb > max_steps) max_steps = nb;
        // every network runs the kernel its own omc_mlp_train_epoch call runs (16-row tiles up to q16_rows rows)
        if (j.batch <= q16_rows) max_batch16 = std::max<int64_t>(max_batch16, j.batch);
        else max_batch32 = std::max<int64_t>(max_batch32, j.batch);
        part_bytes = std::max(part_bytes, omc::mlp_partial_bytes(hidden, layers, j.batch));
        if (j.step + nb > last_step) last_step = j.step + nb;
    }
    if (max_steps > 0x7fffffff) return fail(-3, "too many steps per epoch.");
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t pb = up(part_bytes), wb = up(omc::mlp_wt_bytes(hidden, layers) + 16);
    const size_t lb = up(sizeof(double) * (size_t)n);
    if ((rc = c->mb_slab.ensure(lb + (pb + wb) * (size_t)n))) return rc;
    if ((rc = c->mb_table.ensure(omc::mlp_batch_table_bytes(n)))) return rc;
    // 1 - beta^step for every step this epoch can reach (host libm pow: the numbers the single-problem path uses)
    if (c->mb_beta1 != beta1 || c->mb_beta2 != beta2 || c->mb_bc_cap < (size_t)last_step + 2) {
        const size_t cap = ((size_t)last_step + 2) * 2 + 1024;
        c->mb_bc_host.assign(2 * cap, 0.0);
        for (size_t k = 0; k < cap; ++k) {
            c->mb_bc_host[k] = 1.0 - std::pow(beta1, (double)k);
            c->mb_bc_host[cap + k] = 1.0 - std::pow(beta2, (double)k);
        }
        if ((rc = c->mb_bc.ensure(sizeof(double) * 2 * cap))) return rc;
        HIP_TRY(hipMemcpyAsync(c->mb_bc.p, c->mb_bc_host.data(), sizeof(double) * 2 * cap, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->mb_bc_cap = cap; c->mb_beta1 = beta1; c->mb_beta2 = beta2;
    }
    char* slab = (char*)c->mb_slab.p;
    double* loss = (double*)slab;
    std::vector<omc::MlpBatchJob> hj((size_t)n);
    for (int i = 0; i < n; ++i) {
        omc::MlpBatchJob& b = hj[(size_t)i];
        b.data = jobs[i].data; b.nrows = jobs[i].n_rows; b.batch = jobs[i].batch; b.first_step = jobs[i].step;
        b.params = jobs[i].params; b.adam_m = jobs[i].adam_m; b.adam_v = jobs[i].adam_v;
        b.partial = (float*)(slab + lb + (pb + wb) * (size_t)i);
        b.wt = (float*)(slab + lb + (pb + wb) * (size_t)i + pb);
        b.loss_acc = loss + i;
        b.lr = jobs[i].lr; b.seed = jobs[i].seed; b.shuffle_key = jobs[i].shuffle_key;
        b.allow_q16 = true;
    }
    c->h_table.resize(omc::mlp_batch_table_bytes(n));
    omc::mlp_batch_table_image(hj.data(), n, hidden, layers, beta1, beta2, eps, weight_decay, dropout, c->h_table.data());
    HIP_TRY(hipMemcpyAsync(c->mb_table.p, c->h_table.data(), c->h_table.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(loss, 0, sizeof(double) * (size_t)n, c->stream));
    HIP_TRY(omc::mlp_train_epoch_batch(c->stream, c->mb_table.p, n, hidden, layers, max_steps, (int)((max_batch32 + 31) / 32),
                                       (int)((max_batch16 + 15) / 16),
                                       (const double*)c->mb_bc.p, (const double*)c->mb_bc.p + c->mb_bc_cap));
    c->h_bres.resize((size_t)n);
    HIP_TRY(hipMemcpyAsync(c->h_bres.data(), loss, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) {
        const int64_t nb = (jobs[i].n_rows + jobs[i].batch - 1) / jobs[i].batch;
        jobs[i].step += nb;
        jobs[i].mean_loss = c->h_bres[(size_t)i] / (double)nb;
    }
    return 0;
}

}  // extern "C"
