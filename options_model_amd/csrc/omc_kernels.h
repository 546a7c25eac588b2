// omc_kernels.h -- host-callable launchers of the gfx950 kernels (internal to libomc.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cmath>

namespace omc {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of a kernel ON A DEVICE: a process that prices on
// a second device must set it there too, and host threads with one context each may get here together.  `mask`
// (one static per kernel instantiation) has a bit per device; setting the attribute twice is harmless.
inline hipError_t set_max_dynamic_lds(std::atomic<uint64_t>& mask, const void* kernel, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    mask.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

// ---- omc_paths.hip
hipError_t launch_gbm_paths(hipStream_t st, float* S, int64_t ld, int64_t n_paths, int n_steps,
                            double S0, double r, double sigma, double T, uint64_t seed,
                            uint32_t stream, uint64_t pair_offset, int antithetic, int vec_hint);
hipError_t launch_heston_paths(hipStream_t st, float* S, int64_t ld, int64_t n_paths, int n_steps,
                               double S0, double r, double T, double v0, double kappa,
                               double theta, double xi, double rho, uint64_t seed, uint32_t stream,
                               uint64_t pair_offset, int scheme, int vec_hint);
hipError_t launch_gbm_from_normals(hipStream_t st, float* S, int64_t ld, int64_t n_paths,
                                   int n_steps, double S0, double r, double sigma, double T,
                                   const float* Z, int64_t ldz, int antithetic);
hipError_t launch_heston_from_normals(hipStream_t st, float* S, int64_t ld, int64_t n_paths,
                                      int n_steps, double S0, double r, double T, double v0,
                                      double kappa, double theta, double xi, double rho,
                                      const float* Z1, const float* Z2, int64_t ldz, int scheme);
// calibrator inner loop: terminal spots only (ST device [n_paths]), then one mean per strike
hipError_t launch_heston_terminal_store(hipStream_t st, float* ST, int64_t n_paths, int n_steps,
                                        double S0, double r, double T, double v0, double kappa,
                                        double theta, double xi, double rho, uint64_t seed,
                                        uint32_t stream, uint64_t pair_offset, int scheme);
// out_dev [n_strikes][2] = {sum, sumsq} of max(+-(S_T - K), 0); part_dev: payoff_partial_bytes(n_paths, n_strikes) of scratch
size_t payoff_partial_bytes(int64_t n_paths, int n_quotes);
hipError_t launch_payoff_means(hipStream_t st, const float* ST, int64_t n_paths, const double* K_dev,
                               int n_strikes, int is_put, double* part_dev, double* out_dev);
// a whole quote surface: the expiry on grid.y (its constants and Philox sub-stream from a 64-byte-per-expiry table:
// heston_surface_fill_table makes the host image, the caller copies it), ST device [n_expiries][ldst]; then the quotes
size_t heston_surface_table_bytes(int n_expiries);
void heston_surface_fill_table(void* tab_host, int n_steps, double r, const double* T_host, const uint32_t* stream_host,
                               int n_expiries, double kappa, double theta, double xi, double rho);
hipError_t launch_heston_terminal_surface(hipStream_t st, float* ST, int64_t ldst, int64_t n_paths, int n_steps, double S0,
                                          int n_expiries, double v0, uint64_t seed, uint64_t pair_offset, int scheme,
                                          const void* tab_dev);
hipError_t launch_payoff_means_surface(hipStream_t st, const float* ST, int64_t ldst, int64_t n_paths, const double* K_dev,
                                       const int32_t* expiry_of_dev, int n_quotes, int is_put, double* part_dev, double* out_dev);
hipError_t launch_philox_kat(hipStream_t st, const uint32_t* in, uint32_t* out, int n);
hipError_t launch_gbm_normals(hipStream_t st, float* Z, int64_t ldz, int64_t n_pairs, int n_steps,
                              uint64_t seed, uint32_t stream, uint64_t pair_offset);

constexpr int kMaxLsmBlocks = 1024;  // grid of the per-step sweep (also partial-moment slots)
// discounted terminal payoff partial sums -> part[8][kMaxLsmBlocks] (rows 0,1,3 used)
hipError_t launch_terminal(hipStream_t st, double* part, int* nblk_out, int model, int scheme,
                           int antithetic, int64_t n_paths, int n_steps, double S0, double K,
                           double r, double sigma, double T, double v0, double kappa, double theta,
                           double xi, double rho, int is_put, uint64_t seed, uint32_t stream,
                           uint64_t pair_offset);

// ---- omc_lsm.hip
constexpr int kMaxSteps = 4094;      // discount table / beta table live in LDS

struct LsmProblem {
    const float* S;   // [N+1][ld]
    int64_t ld, M;
    int N, is_put;
    double K, r, T;
    // Antithetic-FOLDED storage (two-pass flow of the fused GBM pricing; omc_lsm_dev.h): non-null = S holds only the FIRST
    // partner of every antithetic pair ([N+1][ld], M / 2 columns) and fold_cK[t] = S0^2 exp(2 drift t) / K (device, N + 1
    // doubles): the partner of a stored spot S_t is C_t / S_t.  M stays the number of PATHS.
    const double* fold_cK = nullptr;
};

// device workspace carved by the API layer (all sizes in elements)
struct LsmWorkspace {
    float* sx;        // [M]   spot at the (current) exercise time of each path
    int32_t* tex;     // [M]   step index of that exercise (N = maturity / never exercised)
    float* live;      // [M]   per-step reference flow: S_N while the path is in play, negative once it has exercised
                      //       (sticky; sx / tex valid where negative)
    double* D;        // [N+1] discount table exp(-r dt k)
    double* part;     // [2][8][kMaxLsmBlocks] per-block partial moments, ping-pong by step parity
    double* gmom;     // [N+1][8] reduced moments per step (row stride `gstride` doubles)
    int gstride = 8;  // K pricings advanced together keep their moments in ONE table [N+1][K][8]: stride 8K
    double* betas;    // [N+1][4] b0,b1,b2,n
    double* part1;    // two-pass: [N+1][ntiles][8] partial moments of pass 1 (one 64-byte record per step and tile)
    int64_t part1_tiles;
    double* result;   // [8] sum, sumsq, n_exercised, n_zero, sum_nitm, -, -, -
    const float* cont = nullptr;  // per-step sweeps, "values" mode: continuation values [N+1][ldc]
    int64_t ldc = 0;
    // optional: events recorded right around the two big kernels of the two-pass flow
    hipEvent_t ev_p1_begin = nullptr, ev_p1_end = nullptr, ev_p2_begin = nullptr, ev_p2_end = nullptr;
};

size_t lsm_part1_tiles(int64_t M);
// LsmProblem::fold_cK: cK[t] = c0 g^t for t = 0 .. N (c0 = S0^2 / K, g = exp(2 drift dt)), N sequential products
hipError_t lsm_fold_table(hipStream_t st, double* cK, int N, double c0, double g);
// its two constants, from the float32 drift exponent and start value the generator itself uses (launch_gbm_paths)
inline void gbm_fold_constants(double S0, double K, double r, double sigma, double T, int n_steps, double* c0, double* g)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    const float a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    const float s0 = (float)S0;
    *c0 = (double)s0 * (double)s0 / K;
    *g = std::exp2(2.0 * (double)a);
}
// part[0][q][0..nblk) -> result[q], result[4] = sum_t gmom[t][0] (N<=1: skipped)
hipError_t lsm_finalize(hipStream_t st, const double* part, const double* gmom, double* result,
                        int nblk, int N);

// per-step flows (semantics 0 reference sticky / 1 textbook).
//   lsm_step(t) for t = N .. 1; when `external_moments` the prologue of step t reads
//   gmom[t] (already reduced, possibly all-reduced across GPUs) instead of the partials.
hipError_t lsm_step(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int semantics,
                    int t, bool external_moments);
// partials of step t -> gmom[t]  (only needed between steps when moments leave the GPU)
hipError_t lsm_reduce_step_moments(hipStream_t st, const LsmWorkspace& w, int t, int nblk);
int lsm_step_blocks(int64_t M);   // grid of pass 2 / valuation sweeps (256-thread blocks)
int lsm_sweep_blocks(int64_t M);  // grid of the per-step sweep
int lsm_step_block_threads();     // its workgroup size (1024; OMC_STEP_BLOCK=512 for experiments)
// The whole per-step sweep with its arguments in a device block (lsm_sweep_args_bytes, filled from
// lsm_sweep_args_image): capturable into a HIP graph whose replays serve every pricing of the same
// geometry (M, N, semantics, ld, alignment of S).
size_t lsm_sweep_args_bytes();
void lsm_sweep_args_image(const LsmProblem& p, const LsmWorkspace& w, int semantics, bool fill_state, void* out,
                          bool external_moments = false);
// K pricings of one geometry per launch (per-step flows): `table_dev` = K images of lsm_sweep_args_image,
// G = lsm_multi_groups(..) workgroups per pricing.  Per pricing the results are those of its own launches.
int lsm_multi_groups(int64_t M, int K, int device_cus);
hipError_t lsm_step_multi(hipStream_t st, const void* table_dev, int K, int G, int semantics, bool vec4, int N, int t);
hipError_t lsm_reduce_step_moments_multi(hipStream_t st, const void* table_dev, int K, int t);
hipError_t lsm_final_multi(hipStream_t st, const void* table_dev, int K, int64_t M);
hipError_t lsm_sweep_indirect(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int semantics,
                              const void* args_dev);

// two-pass flow (semantics 2)
hipError_t lsm_pass1_moments(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w);
// sticky sweep with the betas of w.betas -- or, when solve_from_moments, with fits every workgroup solves itself
// from w.gmom (no dependent solve launch in between; w.betas is still written, by workgroup 0); writes sx/tex
// when write_state, leaves sums in w.result
hipError_t lsm_pass2_apply(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w,
                           bool write_state, bool solve_from_moments = false);

// valuation of (sx,tex): sums into w.result ; tval = 1 (reference flows) or 0 (textbook);
// use_flags: state of the per-step reference sweep (w.live: unexercised paths take (S_N, N)),
// fill_state: also write that into sx / tex
hipError_t lsm_final_reduce(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, int tval,
                            bool use_flags = false, bool fill_state = false);

// ---- continuation-value network of the NN flow (omc_mlp.hip): 7 -> 64 -> 64 (-> 64) -> 1
constexpr int kMlpPartialStride2 = 4800;                   // gradient partial per workgroup, 2 / 3 hidden layers
constexpr int kMlpPartialStride3 = 8960;                   // (>= parameters + 1 loss slot, multiple of 64)
constexpr int kMlpMaxGroups = 256;                         // one workgroup per CU
constexpr int kMlpQ16MaxRows = 4096;                       // minibatches up to here run in 16-row tiles (mlp_train_q16_kernel): one tile per CU
struct MlpTrainPlan {
    const float* data;  // [nrows][8] float32: 7 inputs + target
    float* params;      // [mlp_train_param_count(64, layers)], updated in place
    float* adam_m;
    float* adam_v;
    float* partial;     // mlp_partial_bytes()
    double* loss_acc;   // += batch-mean loss of every step
    int64_t nrows, batch, first_step;  // optimizer steps taken before this call
    int hidden, layers;                // 64 or 128 units x 2 or 3 hidden layers
    float* wt;                         // mlp_wt_bytes(): transposed connections (tile-per-wave trainer)
    double lr, beta1, beta2, eps, weight_decay, dropout;
    uint64_t seed;         // dropout bits
    uint64_t shuffle_key;  // 0: rows in storage order; else a keyed pseudo-random permutation
    bool wt_current = false;  // `wt` already mirrors `params` (the Adam kernel of an earlier call kept it so)
    // minibatches of up to kMlpQ16MaxRows rows may run in 16-row tiles (mlp_train_q16_kernel).  The NN regressor's entry
    // points say yes; the per-step ContNet flow (single and batched, which must agree bit for bit) keeps the 32-row kernel
    bool allow_q16 = false;
    // Sharded training (one rank of a job, omc_mlp_train_epoch_sharded): `data` holds THIS rank's rows of the epoch,
    // already in epoch order; step k trains on rows [step_off[k], step_off[k + 1]) of it -- this rank's part of the
    // global minibatch k, which has min(batch, rows_global - k * batch) rows over all ranks -- scales by the GLOBAL
    // minibatch size, and the gradient sums (+ loss) of all ranks are added through `allreduce` before Adam, so every
    // rank applies the same update.  nrows = this rank's rows; batch = the GLOBAL minibatch size.
    const int64_t* step_off = nullptr;   // HOST, [steps + 1]; null: not sharded
    int64_t rows_global = 0;
    const uint32_t* drop_pos = nullptr;  // device [nrows]: position of row i inside its global minibatch (dropout key)
    double* gred = nullptr;              // device [param count + 1]: the step's reduced gradient + loss sum
    int (*allreduce)(void* user, double* dptr, int count) = nullptr;  // in-place sum over the ranks, stream-ordered
    void* allreduce_user = nullptr;
};
// many small networks of one shape trained side by side: ONE launch pair per optimizer step for all of them (the
// curve entry points train one net per curve point).  Table rows are built on the host (mlp_batch_table_image) and
// live in device memory; bc1 / bc2 are device tables of 1 - beta^step (host libm pow, as the single path uses).
struct MlpBatchJob {
    const float* data;
    int64_t nrows, batch, first_step;
    float *params, *adam_m, *adam_v, *partial, *wt;
    double* loss_acc;
    double lr;
    uint64_t seed, shuffle_key;
    const double* nrows_dev = nullptr;  // non-null: nrows = batch = the double stored there (set size counted on the device)
    bool allow_q16 = false;             // as MlpTrainPlan::allow_q16: the kernel a single omc_mlp_train_epoch call would run
};
size_t mlp_batch_table_bytes(int n);
bool mlp_batch_supported(int hidden, int layers, int64_t batch);
void mlp_batch_table_image(const MlpBatchJob* jobs, int n, int hidden, int layers, double beta1, double beta2, double eps,
                           double weight_decay, double dropout, void* out);
// max_tiles32 / max_tiles16: the largest minibatch among the problems that run in 32-row / in 16-row tiles (0: none)
hipError_t mlp_train_epoch_batch(hipStream_t st, const void* table_dev, int n, int hidden, int layers, int64_t max_steps,
                                 int max_tiles32, int max_tiles16, const double* bc1_dev, const double* bc2_dev);
// rows up to which a minibatch of this width runs in 16-row tiles (0: never; OMC_MLP_Q16 moves it)
int64_t mlp_q16_rows(int hidden);
// one full-batch step for every problem of the table (two hidden layers; the per-step ContNet flow), grid_tiles
// workgroups per problem walking its tiles
// tile_prefix_dev != null: work-list launch -- grid_tiles workgroups IN TOTAL share all problems' tiles evenly
// (prefix sums of the problems' tile counts from mlp_tile_prefix, n + 1 ints, refreshed whenever set sizes change)
hipError_t mlp_train_step_batch(hipStream_t st, const void* table_dev, int n, int hidden, int grid_tiles, int step_base,
                                const double* bc1_dev, const double* bc2_dev, const int* tile_prefix_dev = nullptr);
hipError_t mlp_tile_prefix(hipStream_t st, const void* table_dev, int n, int* prefix_dev);
size_t mlp_partial_bytes(int hidden, int layers, int64_t batch);
size_t mlp_wt_bytes(int hidden, int layers);
int mlp_train_param_count(int hidden, int layers);               // -1: shape not covered by a trainer
int mlp_train_kernel_choice(int hidden, int layers, int64_t batch);  // 0: this batch size is not covered
// keep (1) / drop (0) of every hidden activation as kernel `variant` draws it (0: pass 2, 1 / 2 / 3: the trainers, the
// values of mlp_train_kernel_choice) -> out [layers][n_rows][hidden] (device); keys: per-row dropout key or null (= row)
hipError_t mlp_dropout_masks(hipStream_t st, int variant, int hidden, int layers, int64_t n_rows, const uint32_t* keys,
                             uint32_t step, uint64_t seed, double dropout, uint8_t* out);
// pass 2 of the NN flow: sticky sweep with the network as continuation value -> (sx, tex)
// hidden in {64, 128}, layers (hidden layers) in {2, 3}; mlp_apply_param_count: floats, -1 otherwise
int mlp_apply_param_count(int hidden, int layers);
hipError_t mlp_apply_pass2(hipStream_t st, const LsmProblem& p, int hidden, int layers, const float* params,
                           const double* feat_mean, const double* feat_std, double y_mean, double y_std,
                           double dropout, uint64_t seed, float* sx, int32_t* tex, int64_t col_base0, int64_t col_base1);
// local-vol paths through the implied-vol network (hidden 64, `layers` residual blocks), row f-4
int localvol_param_count(int hidden, int layers);
hipError_t localvol_paths(hipStream_t st, float* S, int64_t ld, int64_t M, int N, int layers, const float* params,
                          const float* Z, double S0, double r, double T, double K, double m_scale,
                          double tau_scale, double eps_out);
// regressor "ols7" (omc_ols7.hip): ONE least-squares fit on the reference's seven features in the two-pass flow
constexpr int kOls7Stats = 36;  // n, mean[7], C[28] (upper triangle, row-major) of [u, u^2, u^3, max(u,0), s, u*s, y], u = x - 1
size_t ols7_scratch_bytes(int64_t M, int N);
hipError_t ols7_comoments(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const double** stats_dev);
hipError_t ols7_pass2(hipStream_t st, const LsmProblem& p, const double* feat_mean, const double* feat_std,
                      const double* w7, double y_mean, double y_std, float* sx, int32_t* tex);
// pass 1 of the NN flow straight from the path matrix (omc_rows.hip): count -> scan -> statistics -> rows
size_t nn_rows_scratch_bytes(int64_t M, int N);
hipError_t nn_rows_count(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const int64_t** total_dev,
                         const double** stats_dev);
hipError_t nn_rows_write(hipStream_t st, const LsmProblem& p, const double* D, void* scratch, const double* feat_mean,
                         const double* feat_std, double y_mean, double y_std, float* data, int64_t cap);
// float64 means / population variances of the regression features and the target over n rows
size_t nn_stats_scratch_bytes();
hipError_t nn_feature_stats(hipStream_t st, const double* x, const int32_t* t, const double* y, int64_t n,
                            double T, double dt, double* scratch, double* out16);
// the permutation the trainer walks (for tests): out[i] = stored row visited at epoch position i
hipError_t mlp_shuffle_indices(hipStream_t st, int64_t n, uint64_t shuffle_key, int64_t* out);
// exclusive prefix of cnt[0..n) into offs[0..n), total into offs[n] (one workgroup; omc_rows.hip)
hipError_t nn_scan_counts(hipStream_t st, const int32_t* cnt, int64_t n, int64_t* offs);
// per time step, the in-the-money paths among columns [0, half) and [half, M) of a path matrix: counts_dev
// [(N - 1)][2] int64, index N - 1 - t (the reference's row order: later steps first)
hipError_t nn_rows_half_counts(hipStream_t st, const LsmProblem& p, int64_t half, int64_t* counts_dev);
// ---- NN regressor sharded over ranks: which rows of an epoch's global minibatches are THIS rank's
// The job's rows in the reference's order (step desc, global column asc) form `nseg` segments, each owned by one
// rank: gstart[s] = global index of its first row (ascending; gstart[nseg] = rows_global), lstart[s] = the row of the
// rank's OWN matrix where it starts, or -1 when another rank owns it (device arrays).  Epoch position i trains global
// row perm(i) (the keyed permutation of the single-GPU trainer over rows_global); this rank's positions, ascending,
// -> sel_row (own row) / sel_i (position).  scratch: mlp_shard_scratch_bytes(rows_global).  *total_dev -> device
// int64 = how many (must equal the rank's row count).
size_t mlp_shard_scratch_bytes(int64_t rows_global);
hipError_t mlp_shard_select(hipStream_t st, int64_t rows_global, uint64_t shuffle_key, const int64_t* gstart,
                            const int64_t* lstart, int nseg, int group /* segments per time step, 0 = unknown */,
                            void* scratch, int64_t* sel_row, int64_t* sel_i, const int64_t** total_dev);
// data_epoch[j] = data[sel_row[j]] (32-byte rows), drop_pos[j] = sel_i[j] mod batch; step_off[k] = first j with
// sel_i[j] >= k * batch for k = 0 .. steps (device int64 [steps + 1])
hipError_t mlp_shard_gather(hipStream_t st, const float* data, const int64_t* sel_row, const int64_t* sel_i,
                            int64_t n_local, int64_t batch, int64_t steps, float* data_epoch, uint32_t* drop_pos,
                            int64_t* step_off);
// one epoch: ceil(nrows / batch) optimizer steps (forward+backward kernel, reduce+Adam kernel)
hipError_t mlp_train_steps(hipStream_t st, const MlpTrainPlan& t);
int64_t mlp_plan_kernel_batch(const MlpTrainPlan& t);  // the minibatch size that picks the kernel / sizes the partials

// ---- omc_contnet.hip: the per-step ContNet(1 -> h -> h -> 1) regressor of the reference's v1 / v2 pricers
int cn_padded_width(int hidden);          // width of the trainer that hosts h units (32 / 64 / 128), -1: too wide
size_t cn_scratch_bytes(int64_t M);
const double* cn_header(const LsmProblem& p, void* scratch);  // device: n, mean, 1/std (1 if std == 0), std
// members of step t's regression set (in the money, not exercised): row offsets and header
hipError_t cn_count(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt);
// the set as trainer rows [xs, 0 x 6, y = payoff(S_N) Dt] in path order; data holds >= n rows of 8 floats
hipError_t cn_rows(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt,
                   float* data);
// nn.Linear default initialisation keyed by (seed, t) in the trainer's parameter layout; Adam moments zeroed
hipError_t cn_init(hipStream_t st, int hidden, int t, uint64_t seed, float* params, float* m, float* v);
// cont[j] = net(xs_j) for the members of step t (float32)
hipError_t cn_forward(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt,
                      int hidden, const float* params, float* cont);

}  // namespace omc
