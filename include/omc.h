/*
 * omc.h -- C ABI of libomc.so: MI355X (gfx950) American-option Monte-Carlo hot path.
 *
 * The reference (Levicoz/Options-model) is 100 % Python and has no FFI, plugin or operator
 * interface; its boundary for this path is the Python call surface.  This header is the
 * drop-in boundary underneath that surface: each entry point names the reference code it
 * replaces (paths relative to the reference root).  Host bindings: ctypes, see
 * options_model_amd/_ffi.py and INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int: 0 ok; <0 invalid argument (Python raises ValueError with
 *     the reference's message); >0 a hipError_t (Python raises RuntimeError).
 *     omc_last_error() returns a thread-local description of the last failure.
 *   - no HIP call happens at load time; a context is created lazily per (process, device),
 *     so the library is safe under the reference's `spawn`ed ProcessPoolExecutor workers
 *     (options_model_2_ui.py:8-11).
 *   - path matrices are float32 [n_steps+1][ld], row = time step, ld >= n_paths elements
 *     (the reference's S[t, j] C-order layout, options_model_3/options_model_3.py:477).
 *     Antithetic partner of column j is j + n_paths/2 (options_model_3.py:476).
 *   - device pointers passed in are borrowed; the library frees only what omc_alloc returned.
 *   - calls on one context are serialised by the caller; all entry points are synchronous
 *     (return after the stream has drained) unless stated otherwise.
 */
#ifndef OMC_H
#define OMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMC_ABI_VERSION 9

typedef struct omc_ctx omc_ctx;

/* semantics of the backward induction */
enum {
    OMC_SEM_REFERENCE = 0, /* per-step sticky flow: Options_model.py:108-157, options_model_2.py:278-313 */
    OMC_SEM_TEXTBOOK = 1,  /* classic Longstaff-Schwartz, discounted to t=0                        */
    OMC_SEM_TWO_PASS = 2   /* v3 flow: options_model_3/options_model_3.py:482-516 + :615-651        */
};
enum { OMC_MODEL_GBM = 0, OMC_MODEL_HESTON = 1 };
enum {
    OMC_HESTON_REFERENCE_CLAMP = 0, /* options_model_3.py:230-233                                 */
    OMC_HESTON_FULL_TRUNCATION = 1, /* Lord et al. (named by BASELINE.json)                       */
    OMC_HESTON_CALIBRATOR = 2       /* heston_calibration.py:242-255: floor 1e-8, arithmetic Euler */
};

typedef struct {
    int32_t model;         /* OMC_MODEL_*                                               */
    int32_t is_put;        /* payoff: options_model_3.py:376-380                          */
    int32_t semantics;     /* OMC_SEM_*                                                 */
    int32_t antithetic;    /* 1 = reference layout (GBM only may pass 0)                */
    int32_t heston_scheme; /* OMC_HESTON_*                                              */
    int32_t n_steps;
    int64_t n_paths;       /* LOCAL paths of this context (even when antithetic)        */
    double S0, K, r, sigma, T;
    double v0, kappa, theta, xi, rho; /* Heston only (dict keys of options_model_2_ui.py:74-80) */
    uint64_t seed;         /* Philox key                                                */
    uint64_t stream;       /* sub-stream id (low 32 bits used): one per pricing call    */
    uint64_t pair_offset;  /* global index of this context's first pair (multi-GPU)     */
} omc_params;

typedef struct {
    double price;          /* mean cash-flow (reference flows: valued at t = dt)   */
    double sum, sumsq;     /* over the local paths                                        */
    double std;            /* population std, as Options_model.py:154                     */
    double zero_prob;      /* P(cash-flow == 0), Options_model.py:155                     */
    int64_t n_paths, n_exercised, n_zero, sum_nitm;
    double ms_paths, ms_lsm, ms_total; /* HIP-event times of this call on the context's stream */
    double ms_pass1, ms_pass2;         /* omc_price_american, two-pass flow: the two big LSM kernels */
    int64_t timed;         /* omc_price_american_seq: 1 = this pricing carried its own HIP events (the ms_* kernel
                              times above are its own), 0 = they repeat those of the latest timed pricing before it */
    int64_t folded;        /* omc_price_american (no S_keep) / _seq: 1 = the pricing ran on antithetic-FOLDED storage
                              (option "fold_antithetic", below): only the first partner of every pair was stored */
} omc_result;

/* ---- library / context --------------------------------------------------------------- */
int omc_abi_version(void);
const char* omc_last_error(void);
int omc_device_count(int* count);
/* Which card a context runs on: its HIP ordinal in this process, its PCI bus id ("0000:c1:00.0", at least 13 bytes + NUL)
 * and its name.  A multi-rank job gathers these to prove that N ranks sit on N DISTINCT cards (bench.py --gpus N fails
 * loudly otherwise).  Any of the output pointers may be NULL. */
int omc_ctx_device_info(omc_ctx* ctx, int* device, char* pci_bus_id, int pci_len, char* name, int name_len);
/* hip_stream == NULL: the context creates its own stream; else it borrows the caller's
 * (e.g. torch.cuda.current_stream().cuda_stream) and never destroys it.
 *
 * Stream ordering of borrowed device pointers (the entry/exit contract):
 *   exit   every entry point returns after the context's stream has drained, so whatever the caller
 *          enqueues afterwards, on any stream, sees the library's writes.
 *   entry  a context that BORROWS a stream is ordered by that stream: the caller's pending work on it
 *          precedes the library's.  A context that OWNS its stream (hip_stream == NULL; created with
 *          hipStreamNonBlocking, so the null stream's implicit synchronisation does not apply to it) orders
 *          itself, on entry of every call that takes a device pointer from the caller, after everything then
 *          pending on the device's DEFAULT (null) stream -- the stream PyTorch queues on unless told otherwise
 *          -- by an event wait; no host blocking.  Work the caller has pending on any OTHER stream (a torch
 *          side stream, its own non-blocking streams) is not ordered: synchronise that stream first, or create
 *          the context on it.  The options_model_amd modules that hand torch tensors to the library create their
 *          contexts on torch's current stream (nn_regressor._ctx_on_torch_stream). */
int omc_ctx_create(int device, void* hip_stream, omc_ctx** out);
int omc_ctx_destroy(omc_ctx* ctx);
int omc_sync(omc_ctx* ctx);
int omc_alloc(omc_ctx* ctx, size_t bytes, void** dptr);
int omc_free(omc_ctx* ctx, void* dptr);
int omc_memcpy_h2d(omc_ctx* ctx, void* dst, const void* src, size_t bytes);
int omc_memcpy_d2h(omc_ctx* ctx, void* dst, const void* src, size_t bytes);
/* knobs: "fold_antithetic" (1 = default: the fused pricing calls -- omc_price_american without S_keep and
 * omc_price_american_seq -- keep antithetic GBM paths of the two-pass flow in FOLDED storage when the pricing has at least
 * 65,536 paths over all ranks (smaller ones are launch-bound and stay on the full matrix, bit-equal to their
 * omc_price_american_batch form); 2 = whatever the size; 0 = never.  Folded storage: since S_t S'_t = S0^2
 * exp(2 drift t) for the partners of a pair, only the first partner's path is generated and stored, half the matrix, and
 * both sweeps price the partner from the same spot through its moneyness (C_t / K) / S_t - 1 in float64 -- same Philox
 * normals, same first partners bit for bit, partner spots equal to the stored float32 ones up to their rounding (prices
 * agree to ~1e-6 relative; omc_result.folded says which storage a pricing used; oracle: orc_lsm_two_pass_folded).
 * Full storage = both partners, the layout omc_gbm_paths + omc_lsm produce),
 * "gbm_vec" / "heston_vec" (pairs per thread: 1,2,4; 0 = auto), "world_size" (ranks behind
 * the all-reduce hook, see below), "step_graph" (1 / 0: replay the per-step sweep as one captured HIP
 * graph or launch its kernels one by one; -1 = default, off: same speed at 1M x 252, and a capture per new
 * geometry costs milliseconds), "seq_overlap" (omc_price_american_seq on a context with a communicator, two-pass flow, equal geometry:
 * 1 = the moment all-reduce of pricing k runs on a second stream under the path generation of pricing k+1
 * (second path buffer) and all result sums travel in one collective at the end -- bit-identical results;
 * 0 = one pricing after the other; -1 = default: off -- a job turns it on after it has checked, on its live
 * communicator, that both forms return the same bits, as bench.py does),
 * "seq_event_stride" (omc_price_american_seq: k > 0 = every k-th pricing of a sequence carries its own HIP
 * events, so a sequence yields several samples of the per-kernel times; 0 = default: the first pricing only),
 * "alloc_limit" (PER PROCESS, bytes; 0 = none: no single buffer of the library -- path matrix, workspace, row scratch --
 * may grow beyond it; a larger request fails like a hipMalloc that found no room, code 2 = hipErrorOutOfMemory.  A
 * budget for a card shared with other tenants; on a distributed context such a rank-local failure is reported on
 * EVERY rank, see "failures only one rank can see" below) */
int omc_set_option(omc_ctx* ctx, const char* key, int64_t value);
/* ---- per-step flows across GPUs without a collective per step ------------------------------ */
/* The per-step flows exchange 8 doubles per pricing after every time step.  Instead of an all-reduce per step, every
 * rank can WRITE its contribution into every peer's memory (xGMI) and sum what arrives itself: omc_p2p_export
 * allocates this rank's mailbox (fine-grained device memory) and returns its 64-byte IPC handle; hand all ranks'
 * handles, in rank order, to omc_p2p_connect on every rank (one node; at most 16 ranks).  From then on a context
 * that is distributed (communicator or hook) runs the per-step exchange as ONE small launch per step -- reduce the
 * rank's partials, publish to all peers, poll the own mailbox (bounded), add the contributions in rank order -- and
 * keeps the collective for everything else (the two-pass flow's moment table, the result sums).  Every rank gets
 * the same bits; a contribution that does not arrive within the deadline makes the pricing call fail (error 3100,
 * sticky: omc_p2p_status reports it) -- it never hangs.  Option "p2p_exchange" = 0 switches back to the collective
 * without disconnecting; "p2p_deadline_ms" = how long an exchange waits for a peer's contribution (default 2000),
 * "p2p_first_deadline_ms" = the same for the FIRST exchange of a pricing call (default 30000: nothing aligns the ranks
 * before it, and a first-use code-object load or a multi-GB allocation on one rank can skew them by seconds).
 * The failure is COLLECTIVE: a rank that gave up poisons its slot in every peer's mailbox and adds its error word to
 * the all-reduced result sums, so every rank of the job returns 3100 -- a peer that was merely slow cannot leave the
 * others with a finite price built on a contribution that was given up on.  */
int omc_p2p_export(omc_ctx* ctx, void* handle_out, size_t bytes /* >= 64 */);
int omc_p2p_connect(omc_ctx* ctx, int rank, int world, const void* handles, size_t bytes /* >= world * 64 */);
int omc_p2p_disconnect(omc_ctx* ctx);
int omc_p2p_status(omc_ctx* ctx, int* connected, int* world, uint64_t* error_word);

/* ---- path generation ------------------------------------------------------------------- */
/* replaces the inline GBM block options_model_3.py:473-480 (== Options_model.py:79-88,
 * options_model_2.py:257-264) and the torch loops option_model_3_gpu.py:117-185 */
int omc_gbm_paths_f32(omc_ctx* ctx, float* S, int64_t ld, int64_t n_paths, int n_steps, double S0,
                      double r, double sigma, double T, uint64_t seed, uint64_t stream,
                      uint64_t pair_offset, int antithetic);
/* replaces simulate_heston_paths_antithetic options_model_3.py:211-251
 * (option_model_3_gpu.py:187-248); the variance never reaches memory */
int omc_heston_paths_f32(omc_ctx* ctx, float* S, int64_t ld, int64_t n_paths, int n_steps,
                         double S0, double r, double T, double v0, double kappa, double theta,
                         double xi, double rho, uint64_t seed, uint64_t stream,
                         uint64_t pair_offset, int scheme);
/* injected-normals parity mode: Zhalf is device float32 [n_steps][ldz], row t-1 drives step t
 * (the exact consumption order of options_model_3.py:475-480 / :223-233) */
int omc_gbm_paths_from_normals_f32(omc_ctx* ctx, float* S, int64_t ld, int64_t n_paths,
                                   int n_steps, double S0, double r, double sigma, double T,
                                   const float* Zhalf, int64_t ldz, int antithetic);
int omc_heston_paths_from_normals_f32(omc_ctx* ctx, float* S, int64_t ld, int64_t n_paths,
                                      int n_steps, double S0, double r, double T, double v0,
                                      double kappa, double theta, double xi, double rho,
                                      const float* Z1half, const float* Z2half, int64_t ldz,
                                      int scheme);
/* RNG taps for known-answer tests: in = n x {ctr[4], key[2]} (host), out = n x 4 (host) */
int omc_philox4x32_10(omc_ctx* ctx, const uint32_t* in, uint32_t* out, int n);
/* the normals the GBM generator consumes: Z device float32 [n_steps][ldz] */
int omc_gbm_normals_f32(omc_ctx* ctx, float* Z, int64_t ldz, int64_t n_pairs, int n_steps,
                        uint64_t seed, uint64_t stream, uint64_t pair_offset);

/* ---- Longstaff-Schwartz backward induction (polynomial regressor) -------------------------- */
/* replaces the backward loops Options_model.py:108-157 / options_model_2.py:278-313
 * (semantics 0), options_model_3.py:482-651 (semantics 2), with the per-step MLP swapped for
 * OLS on [1,u,u^2], u = S/K-1 (the reference accepts lsm_poly_degree and ignores it:
 * Options_model.py:53, options_model_2.py:178-179).
 * betas_out: optional host [n_steps+1][4] = b0,b1,b2,n_itm per step.
 * sx_out / tex_out: optional host [n_paths] final exercise spot / step per path. */
int omc_lsm_poly(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                 double r, double T, int is_put, int semantics, omc_result* res,
                 double* betas_out, float* sx_out, int32_t* tex_out);
/* decision replay with given per-step fits (host betas [n_steps+1][4], n<=0 skips the step) */
int omc_lsm_apply_frozen(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps,
                         double K, double r, double T, int is_put, const double* betas,
                         omc_result* res, float* sx_out, int32_t* tex_out);

/* the per-step flows (semantics 0 / 1) driven by EXTERNALLY supplied continuation values instead of
 * the fitted polynomial: cont is a device float32 matrix [n_steps+1][ldc] (row t = continuation value of
 * every path at step t, the float32 a torch module returns; entries of paths that are out of the money
 * or, under semantics 0, already exercised are never read).  Replays a recorded run of the reference's
 * own per-step loop -- Options_model.py:108-157 / options_model_2.py:278-313, whose ContNet outputs are
 * the `continuation` of :141-142 -- through the kernel that implements its mask, discounting, strict `>`
 * and (mean, std, zero_prob) statistics. */
int omc_lsm_apply_values(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps,
                         double K, double r, double T, int is_put, int semantics, const float* cont,
                         int64_t ldc, omc_result* res, float* sx_out, int32_t* tex_out);

/* The regressor the reference's v1 / v2 pricers run: a FRESH ContNet(1 -> nn_hidden -> nn_hidden -> 1) at
 * every time step (Options_model.py:14-25,112-151; options_model_2.py:283-312, whose constructor arguments
 * nn_hidden / nn_epochs / nn_lr these are): input = the step's regression set (in the money, not yet
 * exercised) standardised by its own mean / population std (std 0: centred only), target = the set's
 * cash-flows valued at the step (not normalised), nn_epochs full-batch Adam(lr = nn_lr) steps on the mean
 * squared error, exercise where payoff > net(input) (strict), sticky mask, cash-flows valued at t = 1.
 * Initialisation is torch's nn.Linear default (uniform +-1/sqrt(fan_in), weights and biases) drawn from
 * Philox keyed by (nn_seed, t): the reference's v1 never seeds torch and its v2 seeds it once per pricing, so
 * individual nets cannot be matched, only the distribution of prices.  nn_hidden 1 .. 128; one GPU (-10 on a
 * context with a communicator or hook).  res->sum_nitm = training rows summed over steps.
 * omc_lsm_contnet works on a device path matrix [n_steps+1][ld]; omc_price_american_contnet generates the
 * paths of `p` first (p->semantics must be OMC_SEM_REFERENCE). */
int omc_lsm_contnet(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, double r,
                    double T, int is_put, int nn_hidden, int nn_epochs, double nn_lr, uint64_t nn_seed,
                    omc_result* res, float* sx_out, int32_t* tex_out);
int omc_price_american_contnet(omc_ctx* ctx, const omc_params* p, int nn_hidden, int nn_epochs, double nn_lr,
                               uint64_t nn_seed, omc_result* res);
/* the initial parameters of step t's net (host float32 [n], n = 8H + H*H + 2H + 1 with H = nn_hidden rounded
 * up to 32 / 64 / 128): layer 0 as [unit][8] = {weight, 0 x 6, bias}, layer 1 as [out][in] then its biases,
 * the output weights, the output bias; entries of units >= nn_hidden are zero.  For tests and for seeding a
 * torch ContNet with the same start. */
int omc_contnet_init_params(omc_ctx* ctx, int nn_hidden, int t, uint64_t nn_seed, float* params_out, int n);

/* multi-GPU: paths shard by antithetic pair, only regression moments and the final sums
 * cross GPUs.  `hook(user, dptr, count)` must all-reduce (sum) `count` DEVICE doubles in place,
 * ordered on the context's stream (RCCL via torch.distributed on the host side).  It is called
 * ONCE with the whole [n_steps+1][8] moment table for the two-pass flow (decision-independent
 * moments), once per time step with 8 doubles for the per-step flows, and finally with the 8
 * result sums {sum, sumsq, n_exercised, n_zero, sum_nitm, ..}.  With a hook installed the
 * returned omc_result therefore carries GLOBAL sums and is normalised by
 * n_paths * world_size (omc_set_option(ctx, "world_size", W); shards are equal). */
typedef int (*omc_allreduce_fn)(void* user, double* dptr, int count);
int omc_set_allreduce_hook(omc_ctx* ctx, omc_allreduce_fn fn, void* user);

/* The same exchange with RCCL called from inside the library (no host callback, no PyTorch):
 * librccl.so is opened on first use.  Rank 0 makes a 128-byte unique id (omc_comm_unique_id) and hands
 * it to the other ranks by any means (options_model_amd/rendezvous.py uses a file on the node); every
 * rank then calls omc_comm_init on its context (collective: ncclCommInitRank on the context's device).
 * From then on the context's pricing calls enqueue ncclAllReduce(sum, double) on its stream for the
 * moment table(s) and the 8 result sums exactly where the hook would have been called, the returned
 * omc_result carries GLOBAL sums, and "world_size" is the communicator's rank count.  The reference has
 * no counterpart (no distributed code: SURVEY.md section 5.8).
 * omc_comm_allreduce_f64: blocking all-reduce of a small HOST vector (op 0 = sum, 1 = max) through the
 * same communicator -- barriers and max-over-ranks timings of a benchmark harness. */
int omc_comm_unique_id(void* uid_out, size_t bytes);
int omc_comm_init(omc_ctx* ctx, int rank, int world, const void* uid, size_t bytes);
int omc_comm_destroy(omc_ctx* ctx);
int omc_comm_info(omc_ctx* ctx, int* rank, int* world); /* world = 0: no communicator */
int omc_comm_allreduce_f64(omc_ctx* ctx, double* host_inout, int count, int op);

/* ---- fused pricing: paths -> LSM -> discounted mean ----------------------------------------- */
/* replaces AdvancedOptionPricer.price_american_enhanced_lsm options_model_3.py:439-651 and
 * price_american_option Options_model.py:44-157 end to end on one GPU.  S_keep: optional
 * caller-owned device matrix [n_steps+1][ld] to receive the paths (NULL: internal workspace) */
int omc_price_american(omc_ctx* ctx, const omc_params* p, omc_result* res, float* S_keep,
                       int64_t ld);
/* European discounted payoff from terminal values only (no path matrix): replaces
 * price_european_streaming options_model_3.py:382-437; sums2 host {sum, sumsq} */
int omc_price_european(omc_ctx* ctx, const omc_params* p, omc_result* res);

/* ---- calibrator inner loop -------------------------------------- */
/* replaces HestonPricer.price_options_batch / price_european_option
 * (options_model_3/heston_calibration.py:259-312) for ONE expiry: simulate n_paths antithetic
 * Heston paths (terminal spots only, no path matrix), then the discounted mean payoff of every
 * strike.  strikes / prices / stderrs are host arrays of n_strikes (stderrs may be NULL). */
int omc_heston_price_strikes(omc_ctx* ctx, int64_t n_paths, int n_steps, double S0, double r,
                             double T, double v0, double kappa, double theta, double xi, double rho,
                             uint64_t seed, uint64_t stream, int scheme, const double* strikes,
                             int n_strikes, int is_put, double* prices, double* stderrs);
/* the same for a whole quote SURFACE -- what ONE evaluation of the calibrator's objective asks for
 * (heston_calibration.py:283-312 loops `for T in unique_T`; :404-472 calls it once per optimizer iteration): n_expiries
 * expiries (host, expiries[e] > 0) each simulated on its own Philox sub-stream streams[e] (host), n_quotes quotes with
 * strike strikes[q] on expiry expiry_of[q] (host int32, 0 .. n_expiries-1).  One launch for all simulations (expiry on
 * grid.y), one for all quotes, one table upload, one read-back, one wait.  Every quote has the bits of its own
 * omc_heston_price_strikes(T = expiries[expiry_of[q]], stream = streams[expiry_of[q]], strike = strikes[q]) call. */
int omc_heston_price_surface(omc_ctx* ctx, int64_t n_paths, int n_steps, double S0, double r, double v0, double kappa,
                             double theta, double xi, double rho, uint64_t seed, int scheme, const double* expiries,
                             const uint64_t* streams, int n_expiries, const double* strikes, const int32_t* expiry_of,
                             int n_quotes, int is_put, double* prices, double* stderrs);

/* ---- NN continuation-value regressor: fused training of the network ------------------------ */
/* replaces the minibatch loop of price_american_enhanced_lsm (options_model_3.py:565-600:
 * SingleLSMNet(7, hidden, layers) :85-103, nn.MSELoss, optim.Adam(lr, weight_decay), shuffled
 * minibatches) for hidden = 32, 64 or 128 with layers = 2 (BASELINE config 5 names 7 -> 64 -> 64 -> 1)
 * or 3 (the depth SingleLSMNet always has in the reference; 3 x 128 is its default) -- see
 * omc_mlp_train_supported for the batch sizes; anything else returns -9.
 * One call = one epoch over `n_rows` rows of
 * `data` ([n_rows][8] float32 device memory: 7 normalised features + normalised target):
 * ceil(n_rows / batch) optimizer steps of float32 MFMA forward/backward + Adam.  The epoch
 * visits the rows in a pseudo-random permutation keyed by `shuffle_key` (a Feistel network with
 * cycle-walking, evaluated in the kernel: no randperm, no gather; 0 = storage order) --
 * omc_mlp_shuffle_indices writes that permutation out (out[i] = row visited at position i).
 * `params` (device, omc_mlp_param_count floats) is laid out W1|b1 as [hidden][8] (bias in
 * column 7), then per further hidden layer W [hidden][hidden] and b [hidden], then the output
 * weights [hidden] and bias [1]; adam_m / adam_v are the moment buffers (zero
 * them before the first epoch); *step counts optimizer steps across calls (bias correction).
 * dropout is applied after each ReLU as in training mode (Philox bits keyed by `seed`).
 * *mean_loss = mean over the epoch's steps of the batch-mean squared error (the value the
 * reference feeds to ReduceLROnPlateau and its early-stopping test, :601-613). */
int omc_mlp_param_count(int hidden, int layers);
/* Pass 2 of the NN flow (options_model_3.py:615-651): sticky backward sweep over a device path
 * matrix with the trained network as continuation value -- features [1, x, x^2, x^3,
 * max(x-1,0), s, x*s] of x = S/K normalised with (feat_mean, feat_std) (host, 7 each), network
 * output scaled back by y_std, y_mean; exercise where payoff > continuation; dropout (> 0) stays
 * active as in the reference, which never calls .eval() on this net; cash-flows valued at
 * t = dt.  Networks: SingleLSMNet(7, hidden, layers) with hidden in {64, 128} and layers in
 * {2, 3} -- the reference's default is (128, 3), options_model_3.py:87 -- anything else: -9.
 * params (device, omc_mlp_param_count(hidden, layers) floats): W1|b1 as [hidden][8] (bias in
 * column 7), then per further hidden layer W [hidden][hidden] and b [hidden], then the output
 * weights [hidden] and bias [1]; for (64, 2) this is omc_mlp_train_epoch's layout.
 * Optional sx_out / tex_out (host). */
int omc_lsm_apply_mlp(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                      double r, double T, int is_put, int hidden, int layers, const float* params,
                      const double* feat_mean, const double* feat_std, double y_mean, double y_std,
                      double dropout, uint64_t seed, omc_result* res, float* sx_out, int32_t* tex_out);
/* the same on one rank's shard of a job: the dropout key of column j is its column in the UNSHARDED matrix,
 * j + col_base0 for the first half of this matrix's columns (first partners) and j - n_paths / 2 + col_base1 for the
 * second -- (pair_offset, P_global + pair_offset) -- so a shard draws the masks the single GPU draws.  The sums in
 * `res` are the shard's; the caller adds them over the ranks. */
int omc_lsm_apply_mlp_shard(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                      double r, double T, int is_put, int hidden, int layers, const float* params,
                      const double* feat_mean, const double* feat_std, double y_mean, double y_std,
                      double dropout, uint64_t seed, omc_result* res, float* sx_out, int32_t* tex_out,
                            int64_t col_base0, int64_t col_base1);
/* Pass 1 of the NN flow (options_model_3.py:482-563) straight from a device path matrix: every
 * in-the-money (step, path), steps N-1 down to 1 and paths ascending within a step (the reference's
 * order), becomes one row of `data` ([rows][8] float32, device): the 7 features [1, x, x^2, x^3,
 * max(x-1,0), s, x*s] of x = S/K, s = sqrt(max(T - t dt, 1e-6)), normalised by their means and
 * population stds over all rows (zero std -> 1), and the target (terminal payoff discounted to t)
 * normalised likewise.  *n_rows = number of rows; with data == NULL only the count is made (call
 * once to size the buffer, then again with data and cap_rows >= *n_rows).  stats16 (host) =
 * feat_mean[7], feat_std[7], y_mean, y_std -- on a context with a communicator / hook those of ALL ranks' rows (see the
 * sharded NN regressor below); every rank of the job must then make the call, also one without any row.  There the
 * call with `data` is COLLECTIVE (two all-reduces of 9 and 8 doubles): a failure only one rank can see (its row buffer
 * too small, no memory for its scratch) travels as a flag in the first of them and EVERY rank returns an error -- the
 * rank's own, 3102 on its peers -- instead of leaving them inside a collective; a job without any in-the-money row
 * returns the default statistics (means 0, stds 1) on every rank.
 * S is read twice IN ALL: one sweep counts the rows of every (step, 256-path tile) and forms the statistics -- power sums
 * around a workgroup's first row (a real row: a column that is constant has deviation 0 exactly), added over its lanes
 * in a fixed order, turned into (n, mean, M2) triples per workgroup and merged by Chan's formula in a fixed two-level
 * tree: the two-pass values of :550-563 to ~1e-14, a constant column's variance exactly 0 -- and one sweep writes the
 * rows (records staged through LDS, contiguous 16-byte stores).  The count call (data == NULL) makes the first sweep and leaves its results in the context; the call with `data`
 * starts from them when it is the NEXT call on this context with the same arguments and S has not been written in
 * between (any other call on the context drops them, and the call with `data` then sweeps again itself). */
int omc_nn_build_rows(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K,
                      double r, double T, int is_put, float* data, int64_t cap_rows, int64_t* n_rows,
                      double* stats16);
/* Regressor "ols7": the v3 two-pass flow (options_model_3.py:482-516 pass 1, :542-563 normalisation, :615-651 pass 2,
 * :651 mean at t = dt) with ONE global least-squares fit on the reference's seven features [1, x, x^2, x^3, max(x-1,0),
 * s, x*s] (create_regression_features, :105-121) in place of the network -- the reference validates `lsm_poly_degree`
 * and never uses it; this is the linear regressor its own features define, between the per-step 3-term
 * polynomial (omc_lsm_poly) and the network.  Pass 1 is one sweep over S (co-moments of the 6 non-constant features and
 * the target -- accumulated for u = x - 1, the same span in a better conditioned basis --, float64, merged by Chan's
 * formula in a fixed order); the fit is lstsq on the normalised design matrix
 * (zero-variance columns get weight 0, as numpy's minimum-norm solution gives them); pass 2 applies it, strict >, sticky.
 * res->sum_nitm = rows of the regression; weights7 (host, may be NULL) = the fit, column 0 the constant; stats16 (host,
 * may be NULL) = feat_mean[7], feat_std[7], y_mean, y_std as omc_nn_build_rows returns them.  On a context with a
 * communicator / hook the call is collective: the fit is over ALL ranks' rows (two all-reduces of 9 and 28 doubles merge the
 * ranks' co-moments, one of 8 the result sums) and `res` is the job's; a failure only one rank can see (no memory for its
 * path matrix -- omc_price_american_ols7's largest allocation -- or workspace, a HIP error in its sweep: the ninth double
 * of the first all-reduce; a HIP error in its pass 2: slot 7 of the result sums, which the kernels leave at zero) travels
 * as a flag and EVERY rank returns an error -- the rank's own, 3103 on its peers -- instead of leaving them inside a
 * collective.  No allocation stands between a rank and a collective its peers have entered: the few hundred bytes the
 * flags travel through exist from omc_comm_init / omc_set_allreduce_hook on. */
int omc_lsm_ols7(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, double r, double T,
                 int is_put, omc_result* res, double* weights7, double* stats16, float* sx_out, int32_t* tex_out);
/* the same as one fused call (paths into the context's own matrix, then omc_lsm_ols7): the facade's regressor="ols7" */
int omc_price_american_ols7(omc_ctx* ctx, const omc_params* p, omc_result* res, double* weights7, double* stats16);
/* Normalisers of the training rows (options_model_3.py:550-563) in float64: for rows i < n_rows
 * with x[i] = S/K, step index t[i] and target y[i] (device arrays), out16[0..6] = means of
 * [x, x^2, x^3, max(x-1,0), s, x*s, y] with s = sqrt(max(T - t*dt, 1e-6)), out16[8..14] =
 * their population variances (two passes: mean first, then squared deviations).  out16: host. */
int omc_nn_feature_stats(omc_ctx* ctx, const double* x, const int32_t* t, const double* y,
                         int64_t n_rows, double T, double dt, double* out16);
/* 1 if omc_mlp_train_epoch covers this network shape at this minibatch size: hidden 32, 64 or 128 (the
 * reference's default width) with 2 or 3 hidden layers, any batch (32 is also the width the per-step ContNet flow
 * trains at, omc_lsm_contnet).  Wider networks (256 units: a connection's operands no longer fit the registers /
 * LDS the kernels keep them in) train and sweep through PyTorch-ROCm, with a RuntimeWarning from nn_regressor. */
int omc_mlp_train_supported(int hidden, int layers, int64_t batch);
int omc_mlp_train_epoch(omc_ctx* ctx, const float* data, int64_t n_rows, int64_t batch, int hidden,
                        int layers, float* params, float* adam_m, float* adam_v, int64_t* step,
                        double lr, double beta1, double beta2, double eps, double weight_decay,
                        double dropout, uint64_t seed, uint64_t shuffle_key, double* mean_loss);
int omc_mlp_shuffle_indices(omc_ctx* ctx, int64_t n_rows, uint64_t shuffle_key, int64_t* out_device);
/* Which of the four trainer kernels omc_mlp_train_epoch runs for this shape at this minibatch size: 1 = one
 * workgroup per 128 rows with the weights in LDS (64 units, more than 32 tiles of 32 rows), 2 = one 32-row tile per
 * wave, 3 = one 32-row tile per workgroup, 4 = one 16-row tile per workgroup (minibatches of up to 4,096 rows: the
 * reference's own min(256, R)), 0 = not covered.  The kernels hold the hidden units in different register orders, so
 * WHICH units dropout drops for a given (seed, step, row) depends on it. */
int omc_mlp_train_variant(int hidden, int layers, int64_t batch);
/* Inspection: the dropout masks themselves.  nn.Dropout's random stream in the reference is torch's global generator
 * (options_model_3.py:455, 85-103); here a unit's 16 random bits come from Philox4x32-10 keyed by `seed` with counter
 * (row key, step, layer / half-tile tag, constant), stretched by a multiply-with-carry stream (csrc/omc_mlp.hip
 * relu_dropout).  out (host, [layers][n_rows][hidden] bytes): 1 = kept, 0 = dropped, as kernel `variant` draws them
 * (0 = pass 2, omc_lsm_apply_mlp: row key = path column, step = time step t; 1 .. 4 = omc_mlp_train_variant: row key
 * = position in the minibatch, step = optimizer step counted from 1).  keys (host, n_rows, may be NULL = 0, 1, 2, ...).
 * oracle/dropout.py restates the definition in numpy; tests compare the two bit for bit and then compare training and
 * pass 2 with the float32 / float64 restatement of the reference's arithmetic under these masks. */
int omc_mlp_dropout_masks(omc_ctx* ctx, int variant, int hidden, int layers, int64_t n_rows, const uint32_t* keys,
                          uint32_t step, uint64_t seed, double dropout, uint8_t* out);

/* ---- the NN regressor sharded over the ranks of a job (options_model_3.py:542-613) ---------
 * The reference trains ONE network on the rows of ALL paths.  With the paths sharded by antithetic pair, every rank
 * builds the rows of its own paths and the job trains the network the single GPU would train:
 *   1. omc_nn_build_rows on a context with a communicator / hook: the normalisers are those of ALL ranks' rows (row
 *      count, sums and squared deviations from the global means all-reduced: 3 x 8 doubles); *n_rows stays the rank's
 *      own count and `data` its own rows.
 *   2. The job's rows in the reference's order (step N-1 .. 1, global column ascending) are the concatenation of
 *      segments (step, half, rank) -- a rank's matrix holds its pairs' first partners in columns [0, P_local) and the
 *      second partners behind them, the global matrix all ranks' first partners, then all second partners.
 *      omc_nn_half_counts returns a rank's segment sizes ([n_steps - 1][2] int64 on the host, index n_steps - 1 - t);
 *      the ranks exchange them (omc_comm_allreduce_f64 of a zero-padded table) and build gstart[nseg + 1] (global index
 *      of each segment's first row, ascending, gstart[nseg] = rows_global) and lstart[nseg] (where the segment starts
 *      among the rank's own rows, -1 for another rank's) -- options_model_amd/nn_dist.py: segment_tables.
 *   3. Per epoch, omc_mlp_shard_epoch evaluates the single-GPU trainer's keyed permutation over rows_global, keeps the
 *      positions whose row this rank owns (ascending), gathers those rows into data_epoch ([n_rows_local][8] float32,
 *      device), their positions inside their minibatch into drop_pos (device uint32: the dropout key, so every rank
 *      draws the masks of the unsharded run) and returns step_off (HOST int64 [steps + 1], steps = ceil(rows_global /
 *      batch)): minibatch k holds this rank's rows [step_off[k], step_off[k + 1]) of data_epoch.
 *   4. omc_mlp_train_epoch_sharded runs the epoch: forward / backward over the rank's part of each minibatch, scaled
 *      by the GLOBAL minibatch size; the gradient sums and the loss sum (parameter count + 1 doubles) are all-reduced
 *      on the context's stream; every rank applies the same Adam step, so the parameters stay identical on all ranks
 *      and equal those of the single-GPU run up to float32 summation order.  *mean_loss is the job's.
 * Pass 2 (omc_lsm_apply_mlp) is local; its sums are added over the ranks by the caller. */
int omc_nn_half_counts(omc_ctx* ctx, const float* S, int64_t ld, int64_t n_paths, int n_steps, double K, int is_put,
                       int64_t* counts /* host [(n_steps - 1)][2] */);
int omc_mlp_shard_epoch(omc_ctx* ctx, const float* data, int64_t n_rows_local, int64_t rows_global, int64_t batch,
                        uint64_t shuffle_key, const int64_t* gstart, const int64_t* lstart, int nseg,
                        int segs_per_step /* 2 x ranks: lets the kernel search the table in two levels; 0 = unknown */,
                        float* data_epoch, uint32_t* drop_pos, int64_t* step_off);
int omc_mlp_train_epoch_sharded(omc_ctx* ctx, const float* data_epoch, int64_t n_rows_local, int64_t rows_global,
                                int64_t batch, int hidden, int layers, float* params, float* adam_m, float* adam_v,
                                int64_t* step, double lr, double beta1, double beta2, double eps, double weight_decay,
                                double dropout, uint64_t seed, const int64_t* step_off, const uint32_t* drop_pos,
                                double* mean_loss);

/* ---- local-vol paths through the implied-vol network ------------------------ */
/* replaces simulate_local_vol_paths_antithetic (options_model_3.py:300-333) together with the
 * IVModel.get_volatility_batch call it makes every step (:263-298): S_t = S_{t-1} exp((r -
 * sigma^2/2) dt + sigma sqrt(dt) z) with sigma = ImprovedIVNetwork(log(K/S)/m_scale, tau/tau_scale)
 * (NN_training_stock_iv.py:109-155; hidden_dim 64, `layers` residual LayerNorm/GELU blocks),
 * clamped at `epsilon` and 1e-6, evaluated inside the kernel on the matrix cores.  S: device
 * [n_steps+1][ld] float32, antithetic partner of column j is j + n_paths/2; Z: device
 * [n_steps][n_paths/2] float32 normals (omc_gbm_normals_f32 makes them, or inject your own).
 * params (device, omc_localvol_param_count floats): input_proj as [64][4] (w_m, w_tau, bias, 0),
 * per block Linear W [64][64], b [64], LayerNorm gamma [64], beta [64], then output w [64], b. */
int omc_localvol_param_count(int hidden, int layers);
int omc_localvol_paths_f32(omc_ctx* ctx, float* S, int64_t ld, int64_t n_paths, int n_steps, double S0,
                           double r, double T, double K, int hidden, int layers, const float* params,
                           double m_scale, double tau_scale, double epsilon, const float* Z);

/* ---- many small networks trained side by side (curves with the NN regressor) --------------------------------- */
/* The reference prices a value-vs-expiry curve point by point, training a fresh SingleLSMNet per point
 * (compute_curve_for_S0, options_model_3.py:697-713 -> :565-613); at its minibatch of 256 rows one network occupies
 * 8 of the chip's 256 CUs.  This trains n networks of ONE shape (hidden x layers) for one epoch each, side by side:
 * one launch pair per optimizer step for all of them (problem index on the grid), each network on its own rows,
 * learning rate, dropout / shuffle keys and step counter.  Every network ends the epoch with exactly the parameters,
 * Adam moments and mean loss that its own omc_mlp_train_epoch call produces (same kernel body, same reductions).
 * jobs[i].step is advanced by the epoch's steps, jobs[i].mean_loss receives the epoch-mean batch loss.  Pointers are
 * device pointers as in omc_mlp_train_epoch.  Shapes: 32 | 64 | 128 units x 2 | 3 hidden layers at
 * minibatches of at most 8192 rows. */
typedef struct {
    const float* data;     /* [n_rows][8] float32: 7 normalised features + normalised target        */
    int64_t n_rows, batch;
    float* params;         /* omc_mlp_param_count(hidden, layers) floats, updated in place           */
    float* adam_m;
    float* adam_v;
    int64_t step;          /* in: optimizer steps taken so far; out: + this epoch's                  */
    double lr;
    uint64_t seed;         /* dropout bits                                                          */
    uint64_t shuffle_key;  /* 0: storage order; else this epoch's keyed permutation                 */
    double mean_loss;      /* out                                                                   */
} omc_mlp_job;
int omc_mlp_train_batch_supported(int hidden, int layers, int64_t batch); /* 1: this shape / minibatch is covered */
int omc_mlp_train_epoch_batch(omc_ctx* ctx, omc_mlp_job* jobs, int n, int hidden, int layers, double beta1,
                              double beta2, double eps, double weight_decay, double dropout);

/* ---- a sequence of pricings without host synchronisation in between ------------------------- */
/* n independent pricings enqueued back to back on the context's stream (pricing i + 1 is launched while
 * pricing i runs; every pricing's result sums land in their own slot of a host-mapped buffer; one wait
 * at the end).  res[i] equals what omc_price_american(p[i]) returns, bit for bit; ms_paths / ms_pass1 /
 * ms_pass2 are measured on the first pricing (and on every k-th one with option "seq_event_stride" = k;
 * res[i].timed marks them), ms_total is the average over the sequence.  Across GPUs the
 * moment tables are all-reduced per pricing as usual, but the result sums of all n pricings travel in ONE
 * collective of 8n doubles after the last pricing (the hook is called once with count = 8n); a hook must only
 * ENQUEUE its collective on the stream (as torch.distributed does), then the sequence stays free of host
 * waits across ranks too.  With a native communicator see also option "seq_overlap". */
int omc_price_american_seq(omc_ctx* ctx, const omc_params* p, int n, omc_result* res);
/* Per-step flows (semantics 0 / 1; the kernel of Options_model.py:108-157): a sequence whose pricings share
 * (n_paths, n_steps, r, T, semantics) advances K of them with EVERY launch of the per-timestep kernel -- K path
 * matrices resident, one launch boundary per time step for all K, and across GPUs the K moment vectors of a step in
 * ONE all-reduce of 8K doubles.  One pricing alone is latency-bound (13 MB and ~6 us per launch at 1M paths); K of
 * them fill the chip.  res[i] still carries the bits of omc_price_american(p[i]): the summation tree of a pricing
 * does not depend on K.  Option "seq_step_k": -1 = default (as many as keep one launch within ~200 MB, at most 32:
 * 16 at 1M paths, 32 at 250k), 1 = off, k <= 32; K is further limited by a byte
 * budget for the resident matrices (OMC_SEQ_STEP_BYTES, default 64e9).  This returns the K a sequence would use
 * (1 = one pricing at a time, 0 = invalid arguments). */
int omc_seq_step_width(omc_ctx* ctx, const omc_params* p, int n);

/* ---- many small pricings in one go ------------------------------------------------------- */
/* replaces the curve loops compute_curve_for_S0 (options_model_3.py:697-713, Options_model.py:
 * 190-211, options_model_2.py:336-355) and their ProcessPoolExecutor fan-out: n independent
 * problems (each its own S0, T, n_steps, n_paths, seed ...) run as ONE set of launches with
 * the batch index on the grid.  All problems of a call share model, semantics, antithetic and
 * Heston scheme; results are identical to n calls of omc_price_american / omc_price_european -- bit for bit when the
 * batch runs the kernels a single call runs: the 16-byte ones, taken when EVERY member has whole groups of four path
 * pairs (n_paths % 8 == 0 with antithetic pairs, % 4 without); otherwise all members run the scalar kernels, whose sums
 * are formed in another block geometry: the same decisions (exact ties of a handful of paths aside), prices to 1e-12.
 * Members are priced on FULL path storage: a member large enough for a single call to fold it (65,536+ paths, option
 * "fold_antithetic") equals the single call made with that option at 0.
 * Whole-batch kernel times are reported in res[0]. */
int omc_price_american_batch(omc_ctx* ctx, const omc_params* p, int n, omc_result* res);
int omc_price_european_batch(omc_ctx* ctx, const omc_params* p, int n, omc_result* res);
/* The same for the regressor the v1 / v2 pricers really run (omc_price_american_contnet: a fresh ContNet per time
 * step): n pricings, problem index on the grid for the whole chain (regression set -> rows -> fresh net -> nn_epochs
 * full-batch Adam steps -> continuation values -> decision), the set sizes never leave the device, no host
 * read-back between the first and the last launch.  nn_seeds[i] keys problem i's nets.  res[i] equals
 * omc_price_american_contnet(p[i], ..., nn_seeds[i]) bit for bit. */
int omc_price_american_contnet_batch(omc_ctx* ctx, const omc_params* p, int n, int nn_hidden, int nn_epochs,
                                     double nn_lr, const uint64_t* nn_seeds, omc_result* res);

#ifdef __cplusplus
}
#endif
#endif /* OMC_H */
