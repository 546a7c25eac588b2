/*
 * omc_oracle.c -- CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (options_model_amd/) never links, imports or calls it.
 *
 * What it restates (reference = /root/reference, 100 % Python; citations are file:line):
 *   - GBM exact log-normal recurrence with antithetic pairs (partner of column j is
 *     j + M/2, rows are time):              options_model_3/options_model_3.py:473-480
 *   - Heston Euler with clamp-at-store:     options_model_3/options_model_3.py:211-251
 *     (scheme 1 = Lord et al. full truncation is an extension named by BASELINE.json)
 *   - payoff                                options_model_3/options_model_3.py:376-380
 *   - backward sweeps: per-step sticky flow Options_model.py:108-157 /
 *     options_model_2.py:278-313, two-pass flow options_model_3.py:482-516,615-651,
 *     with the per-step MLP replaced by OLS on [1,u,u^2], u = S/K - 1 (the reference has
 *     no polynomial regressor: SURVEY.md F1; the flows themselves are pinned by the
 *     golden fixtures tests/golden/poly_flows.npz made by tools/capture_golden.py).
 *   - global 7-feature OLS regressor on the reference features
 *                                           options_model_3/options_model_3.py:105-121,550-563
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "Numerics"):
 *   paths are float32, every sum / solve / discount / comparison is float64.
 *   RNG = Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 known answers are in
 *   tests/golden/scalars.json) + Box-Muller.  numpy's PCG64+ziggurat stream used by the
 *   reference is not reproduced; identical-normals parity goes through the
 *   *_from_normals entry points.
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp); path loops and the LSM sweeps
 * are OpenMP-parallel over paths (static schedule: sums are reproducible for a given thread count).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_TWO_PI 6.283185307179586476925286766559

/* ---------------------------------------------------------------- Philox4x32-10 */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* two uint32 -> two standard normals (float32 contract, evaluated via libm double) */
static inline void box_muller(uint32_t a, uint32_t b, float *zc, float *zs)
{
    float u1 = fmaf((float)(a >> 8), 0x1p-24f, 0x1p-25f); /* (0,1] */
    float u2 = (float)(b >> 8) * 0x1p-24f;                /* [0,1) */
    float rad = (float)sqrt(-2.0 * log((double)u1));
    double ang = ORC_TWO_PI * (double)u2;
    *zc = rad * (float)cos(ang);
    *zs = rad * (float)sin(ang);
}

/* 4 normals of one Philox block: counter = (pair_lo, pair_hi, block, stream), key = seed */
void orc_normals4(uint64_t seed, uint64_t pair, uint32_t block, uint32_t stream, float z[4])
{
    uint32_t ctr[4] = {(uint32_t)pair, (uint32_t)(pair >> 32), block, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    box_muller(o[0], o[1], &z[0], &z[1]);
    box_muller(o[2], o[3], &z[2], &z[3]);
}

/* dump normals as the GBM generator consumes them: Z[t-1][p], t=1..n_steps (4 steps/block) */
void orc_gbm_normals_f32(float *Z, int64_t ldz, int64_t n_pairs, int n_steps, uint64_t seed,
                         uint32_t stream, uint64_t pair_offset)
{
#pragma omp parallel for schedule(static) if (n_pairs * (int64_t)n_steps >= 1000000)
    for (int64_t p = 0; p < n_pairs; ++p) {
        float z[4];
        for (int t = 0; t < n_steps; ++t) {
            if ((t & 3) == 0) orc_normals4(seed, pair_offset + (uint64_t)p, (uint32_t)(t >> 2), stream, z);
            Z[(int64_t)t * ldz + p] = z[t & 3];
        }
    }
}

/* ---------------------------------------------------------------- GBM paths */
/* S is [n_steps+1][ld] float32; antithetic: n_paths even, pair p -> columns p and p+P. */
static inline float exp2f_ref(float a) { return (float)exp2((double)a); }

void orc_gbm_paths_f32(float *S, int64_t ld, int64_t n_paths, int n_steps, double S0, double r,
                       double sigma, double T, uint64_t seed, uint32_t stream,
                       uint64_t pair_offset, int antithetic)
{
    const double dt = T / n_steps;
    const double L2E = 1.4426950408889634074;
    const float a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    const float b = (float)(sigma * sqrt(dt) * L2E);
    const int64_t P = antithetic ? n_paths / 2 : n_paths;
#pragma omp parallel for schedule(static) if (P * (int64_t)n_steps >= 1000000)
    for (int64_t p = 0; p < P; ++p) {
        float s0 = (float)S0, s1 = (float)S0, z[4];
        S[p] = s0;
        if (antithetic) S[p + P] = s1;
        for (int t = 1; t <= n_steps; ++t) {
            int i = (t - 1) & 3;
            if (i == 0) orc_normals4(seed, pair_offset + (uint64_t)p, (uint32_t)((t - 1) >> 2), stream, z);
            s0 = s0 * exp2f_ref(fmaf(b, z[i], a));
            S[(int64_t)t * ld + p] = s0;
            if (antithetic) {
                s1 = s1 * exp2f_ref(fmaf(-b, z[i], a));
                S[(int64_t)t * ld + p + P] = s1;
            }
        }
    }
}

/* injected normals: Zhalf is [n_steps][ldz] float32, row t-1 drives step t
 * (options_model_3.py:475-480) */
void orc_gbm_paths_from_normals_f32(float *S, int64_t ld, int64_t n_paths, int n_steps, double S0,
                                    double r, double sigma, double T, const float *Zhalf,
                                    int64_t ldz, int antithetic)
{
    const double dt = T / n_steps;
    const double L2E = 1.4426950408889634074;
    const float a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    const float b = (float)(sigma * sqrt(dt) * L2E);
    const int64_t P = antithetic ? n_paths / 2 : n_paths;
#pragma omp parallel for schedule(static) if (P * (int64_t)n_steps >= 1000000)
    for (int64_t p = 0; p < P; ++p) {
        float s0 = (float)S0, s1 = (float)S0;
        S[p] = s0;
        if (antithetic) S[p + P] = s1;
        for (int t = 1; t <= n_steps; ++t) {
            float z = Zhalf[(int64_t)(t - 1) * ldz + p];
            s0 = s0 * exp2f_ref(fmaf(b, z, a));
            S[(int64_t)t * ld + p] = s0;
            if (antithetic) {
                s1 = s1 * exp2f_ref(fmaf(-b, z, a));
                S[(int64_t)t * ld + p + P] = s1;
            }
        }
    }
}

/* ---------------------------------------------------------------- Heston paths */
typedef struct {
    float dtf, kdt, theta, xi, rho, rho2, rdt_l2, hdt_l2, l2e;
    float sqdt, rdt, xi_sqdt; /* sqrt(dt), r*dt, xi*sqrt(dt) */
    float l2e_sqdt;           /* log2(e)*sqrt(dt) */
} heston_consts;

static heston_consts heston_make(double r, double T, int n_steps, double kappa, double theta,
                                 double xi, double rho)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    heston_consts c;
    c.dtf = (float)dt;
    c.kdt = (float)(kappa * dt);
    c.theta = (float)theta;
    c.xi = (float)xi;
    c.rho = (float)rho;
    c.rho2 = (float)sqrt(1.0 - rho * rho);
    c.rdt_l2 = (float)(r * dt * L2E);
    c.hdt_l2 = (float)(0.5 * dt * L2E);
    c.l2e = (float)L2E;
    c.sqdt = (float)sqrt(dt);
    c.rdt = (float)(r * dt);
    c.xi_sqdt = (float)(xi * sqrt(dt));
    c.l2e_sqdt = (float)(L2E * sqrt(dt));
    return c;
}

/* one Euler step; scheme 0 = reference clamp-at-store (options_model_3.py:230-233),
 * scheme 1 = full truncation (v carried unclamped). */
static inline void heston_step(const heston_consts *c, int scheme, float z1, float z2, float *s,
                               float *v)
{
    if (scheme == 2) { /* options_model_3/heston_calibration.py:242-255 */
        float vp = fmaxf(*v, 1e-8f);
        float sq = sqrtf(vp);
        float w2 = fmaf(c->rho, z1, c->rho2 * z2);
        float vn = fmaf(c->xi_sqdt * sq, w2, fmaf(c->kdt, c->theta - vp, vp));
        *s = fmaf(*s, fmaf(sq * c->sqdt, z1, c->rdt), *s);
        *v = fmaxf(vn, 1e-8f);
        return;
    }
    /* float32 operation order shared with the HIP kernels (heston_pair_step): the pair-level
     * products tw = xi sqrt(dt) w2 and az = log2(e) sqrt(dt) z1 first (the antithetic partner's are
     * their exact negations, which calling this with (-z1, -z2) reproduces bit for bit), then per path */
    float vp = fmaxf(*v, 0.0f);
    float sq = sqrtf(vp);
    float w2 = fmaf(c->rho, z1, c->rho2 * z2);
    float tw = c->xi_sqdt * w2;
    float az = c->l2e_sqdt * z1;
    float base = scheme ? *v : vp;
    float vn = fmaf(sq, tw, fmaf(c->kdt, c->theta - vp, base));
    float arg = fmaf(sq, az, fmaf(-c->hdt_l2, vp, c->rdt_l2));
    *s = *s * exp2f_ref(arg);
    *v = scheme ? vn : fmaxf(vn, 0.0f);
}

/* Philox: one block per pair per 2 steps: (z1,z2) of step 2k+1 from words 0,1; step 2k+2 from 2,3 */
void orc_heston_paths_f32(float *S, int64_t ld, int64_t n_paths, int n_steps, double S0, double r,
                          double T, double v0, double kappa, double theta, double xi, double rho,
                          uint64_t seed, uint32_t stream, uint64_t pair_offset, int scheme)
{
    const heston_consts c = heston_make(r, T, n_steps, kappa, theta, xi, rho);
    const int64_t P = n_paths / 2;
#pragma omp parallel for schedule(static) if (P * (int64_t)n_steps >= 1000000)
    for (int64_t p = 0; p < P; ++p) {
        float s0 = (float)S0, s1 = (float)S0, va = (float)v0, vb = (float)v0, z[4];
        S[p] = s0;
        S[p + P] = s1;
        for (int t = 1; t <= n_steps; ++t) {
            int i = (t - 1) & 1;
            if (i == 0) orc_normals4(seed, pair_offset + (uint64_t)p, (uint32_t)((t - 1) >> 1), stream, z);
            heston_step(&c, scheme, z[2 * i], z[2 * i + 1], &s0, &va);
            heston_step(&c, scheme, -z[2 * i], -z[2 * i + 1], &s1, &vb);
            S[(int64_t)t * ld + p] = s0;
            S[(int64_t)t * ld + p + P] = s1;
        }
    }
}

void orc_heston_paths_from_normals_f32(float *S, int64_t ld, int64_t n_paths, int n_steps,
                                       double S0, double r, double T, double v0, double kappa,
                                       double theta, double xi, double rho, const float *Z1,
                                       const float *Z2, int64_t ldz, int scheme)
{
    const heston_consts c = heston_make(r, T, n_steps, kappa, theta, xi, rho);
    const int64_t P = n_paths / 2;
#pragma omp parallel for schedule(static) if (P * (int64_t)n_steps >= 1000000)
    for (int64_t p = 0; p < P; ++p) {
        float s0 = (float)S0, s1 = (float)S0, va = (float)v0, vb = (float)v0;
        S[p] = s0;
        S[p + P] = s1;
        for (int t = 1; t <= n_steps; ++t) {
            float z1 = Z1[(int64_t)(t - 1) * ldz + p], z2 = Z2[(int64_t)(t - 1) * ldz + p];
            heston_step(&c, scheme, z1, z2, &s0, &va);
            heston_step(&c, scheme, -z1, -z2, &s1, &vb);
            S[(int64_t)t * ld + p] = s0;
            S[(int64_t)t * ld + p + P] = s1;
        }
    }
}

/* ---------------------------------------------------------------- LSM (polynomial) */
typedef struct {
    double price, sum, sumsq;     /* sum / sumsq of the valued cash-flows           */
    int64_t n_paths, n_exercised; /* exercised before maturity                      */
    int64_t n_zero;               /* cash-flow == 0 (Options_model.py:155)          */
    int64_t sum_nitm;             /* sum over t of regression-set sizes             */
} orc_lsm_result;

/* OLS of y on [1,u,u^2] from the 8 sums m = {n,Su,Su2,Su3,Su4,Sy,Suy,Su2y}.
 * Degree is reduced to n-1 for n<3 and whenever a pivot of the LDL^T factorisation is
 * not safely positive (rank-deficient regression set). */
static void solve_poly2(const double m[8], double beta[3])
{
    const double n = m[0];
    beta[0] = beta[1] = beta[2] = 0.0;
    if (n < 0.5) return;
    const double mu = m[1] / n, my = m[5] / n;
    /* centred second-order quantities (all sums over the set) */
    const double c11 = m[2] - m[1] * mu;              /* S(u-mu)^2             */
    const double c1y = m[6] - m[1] * my;              /* S(u-mu)(y-my)         */
    if (n < 1.5 || !(c11 > 1e-14 * fabs(m[2]) + 1e-300)) { beta[0] = my; return; }
    /* q = u^2 centred: S(q-mq)^2, S(u-mu)(q-mq), S(q-mq)(y-my) */
    const double mq = m[2] / n;
    const double c22 = m[4] - m[2] * mq;
    const double c12 = m[3] - m[1] * mq;
    const double c2y = m[7] - m[2] * my;
    const double l21 = c12 / c11;
    const double d2 = c22 - l21 * c12;
    if (n < 2.5 || !(d2 > 1e-12 * fabs(c22) + 1e-300)) {
        beta[1] = c1y / c11;
        beta[0] = my - beta[1] * mu;
        return;
    }
    const double b2 = (c2y - l21 * c1y) / d2;
    const double b1 = (c1y - c12 * b2) / c11;
    beta[2] = b2;
    beta[1] = b1;
    beta[0] = my - b1 * mu - b2 * mq;
}

static inline double payoff_d(double s, double K, int is_put) { return is_put ? K - s : s - K; }

/* semantics: 0 reference per-step sticky, 1 textbook, 2 reference two-pass (v3 flow).
 * betas_out (optional) [n_steps+1][3], nitm_out (optional) [n_steps+1],
 * sx_out / tex_out (optional) final per-path exercise spot and step. */
int orc_lsm_poly(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K, double r,
                 double T, int is_put, int semantics, orc_lsm_result *res, double *betas_out,
                 int64_t *nitm_out, float *sx_out, int32_t *tex_out)
{
    const int N = n_steps;
    const int64_t M = n_paths;
    const double dt = T / N, invK = 1.0 / K;
    double *D = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    float *sx = (float *)malloc(sizeof(float) * (size_t)M);
    int32_t *tex = (int32_t *)malloc(sizeof(int32_t) * (size_t)M);
    double *betas = (double *)calloc((size_t)(N + 1) * 3, sizeof(double));
    int64_t *nitm = (int64_t *)calloc((size_t)(N + 1), sizeof(int64_t));
    if (!D || !sx || !tex || !betas || !nitm) return -1;
    for (int k = 0; k <= N; ++k) D[k] = exp(-r * dt * (double)k);
    const float *SN = S + (int64_t)N * ld;
    for (int64_t j = 0; j < M; ++j) { sx[j] = SN[j]; tex[j] = N; }

    if (semantics == 2) { /* pass 1: targets are discounted terminal payoffs, no decisions */
        for (int t = N - 1; t >= 1; --t) {
            const float *St = S + (int64_t)t * ld;
            double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
#pragma omp parallel for schedule(static) reduction(+ : m0, m1, m2, m3, m4, m5, m6, m7) if (M >= 262144)
            for (int64_t j = 0; j < M; ++j) {
                double imm = payoff_d((double)St[j], K, is_put);
                if (!(imm > 0.0)) continue;
                double pN = payoff_d((double)SN[j], K, is_put);
                double y = (pN > 0.0 ? pN : 0.0) * D[N - t];
                double u = fma((double)St[j], invK, -1.0), u2 = u * u;
                m0 += 1.0; m1 += u; m2 += u2; m3 += u2 * u; m4 += u2 * u2;
                m5 += y; m6 += u * y; m7 += u2 * y;
            }
            double m[8] = {m0, m1, m2, m3, m4, m5, m6, m7};
            nitm[t] = (int64_t)m[0];
            solve_poly2(m, betas + 3 * t);
        }
    }
    for (int t = N - 1; t >= 1; --t) {
        const float *St = S + (int64_t)t * ld;
        if (semantics != 2) {
            double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
#pragma omp parallel for schedule(static) reduction(+ : m0, m1, m2, m3, m4, m5, m6, m7) if (M >= 262144)
            for (int64_t j = 0; j < M; ++j) {
                double imm = payoff_d((double)St[j], K, is_put);
                if (!(imm > 0.0)) continue;
                if (semantics == 0 && tex[j] < N) continue;
                double p = payoff_d((double)sx[j], K, is_put);
                double y = (p > 0.0 ? p : 0.0) * D[tex[j] - t];
                double u = fma((double)St[j], invK, -1.0), u2 = u * u;
                m0 += 1.0; m1 += u; m2 += u2; m3 += u2 * u; m4 += u2 * u2;
                m5 += y; m6 += u * y; m7 += u2 * y;
            }
            double m[8] = {m0, m1, m2, m3, m4, m5, m6, m7};
            nitm[t] = (int64_t)m[0];
            solve_poly2(m, betas + 3 * t);
        }
        if (nitm[t] == 0) continue;
        const double *b = betas + 3 * t;
#pragma omp parallel for schedule(static) if (M >= 262144)
        for (int64_t j = 0; j < M; ++j) {
            double imm = payoff_d((double)St[j], K, is_put);
            if (!(imm > 0.0)) continue;
            if (semantics != 1 && tex[j] < N) continue;
            double u = fma((double)St[j], invK, -1.0);
            double cont = fma(u, fma(u, b[2], b[1]), b[0]);
            if (imm > cont) { sx[j] = St[j]; tex[j] = t; }
        }
    }
    /* valuation: reference flows stop at t=1 without the last discount */
    const int tval = (semantics == 1) ? 0 : 1;
    double sum = 0.0, sumsq = 0.0;
    int64_t nex = 0, nzero = 0, snitm = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum, sumsq, nex, nzero) if (M >= 262144)
    for (int64_t j = 0; j < M; ++j) {
        double p = payoff_d((double)sx[j], K, is_put);
        double cf = (p > 0.0 ? p : 0.0) * D[tex[j] - tval];
        sum += cf; sumsq += cf * cf;
        nex += tex[j] < N;
        nzero += cf == 0.0;
    }
    for (int t = 1; t < N; ++t) snitm += nitm[t];
    if (res) {
        res->sum = sum; res->sumsq = sumsq; res->price = sum / (double)M;
        res->n_paths = M; res->n_exercised = nex; res->n_zero = nzero; res->sum_nitm = snitm;
    }
    if (betas_out) memcpy(betas_out, betas, sizeof(double) * (size_t)(N + 1) * 3);
    if (nitm_out) memcpy(nitm_out, nitm, sizeof(int64_t) * (size_t)(N + 1));
    if (sx_out) memcpy(sx_out, sx, sizeof(float) * (size_t)M);
    if (tex_out) memcpy(tex_out, tex, sizeof(int32_t) * (size_t)M);
    free(D); free(sx); free(tex); free(betas); free(nitm);
    return 0;
}

/* sticky pass 2 with given per-step betas (decision replay for fixtures) */
int orc_lsm_apply_frozen(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K,
                         double r, double T, int is_put, const double *betas,
                         const int64_t *nitm, orc_lsm_result *res, float *sx_out, int32_t *tex_out)
{
    const int N = n_steps;
    const int64_t M = n_paths;
    const double dt = T / N, invK = 1.0 / K;
    double sum = 0.0, sumsq = 0.0;
    int64_t nex = 0, nzero = 0;
    for (int64_t j = 0; j < M; ++j) {
        float sx = S[(int64_t)N * ld + j];
        int tex = N;
        for (int t = N - 1; t >= 1; --t) {
            if (nitm && nitm[t] == 0) continue;
            float st = S[(int64_t)t * ld + j];
            double imm = payoff_d((double)st, K, is_put);
            if (!(imm > 0.0)) continue;
            const double *b = betas + 3 * t;
            double u = fma((double)st, invK, -1.0);
            double cont = fma(u, fma(u, b[2], b[1]), b[0]);
            if (imm > cont) { sx = st; tex = t; break; }
        }
        double p = payoff_d((double)sx, K, is_put);
        double cf = (p > 0.0 ? p : 0.0) * exp(-r * dt * (double)(tex - 1));
        sum += cf; sumsq += cf * cf; nex += tex < N; nzero += cf == 0.0;
        if (sx_out) sx_out[j] = sx;
        if (tex_out) tex_out[j] = tex;
    }
    if (res) {
        res->sum = sum; res->sumsq = sumsq; res->price = sum / (double)M; res->n_paths = M;
        res->n_exercised = nex; res->n_zero = nzero; res->sum_nitm = 0;
    }
    return 0;
}

/* ---- the two-pass flow on antithetic-FOLDED storage (product: lsm_pass1_fold_body / lsm_pass2_fold_body in
 * options_model_amd/csrc/omc_lsm_dev.h).  For GBM the partners of an antithetic pair (options_model_3.py:473-480: the
 * second half of the matrix is driven by -Z) satisfy S_t S'_t = S0^2 exp(2 drift t) =: C_t, so only the first partner
 * is stored (S: [n_steps+1][ld], n_pairs columns) and the partner enters through its moneyness
 *      u' = (C_t / K) / S_t - 1 = fma(cK[t], 1 / S_t, -1)          cK[t] = c0 g^t by sequential products,
 * payoff -K u' (put) / K u' (call), in the money <=> that is > 0.  Everything else is orc_lsm_poly's semantics 2
 * (options_model_3.py:482-516 pass 1, :615-651 sticky pass 2, valued at t = dt) over the 2 n_pairs paths.
 * texa_out / texb_out (optional): exercise step of the stored path / of its partner. */
static inline double fold_pay(double u, double K, int is_put) { return is_put ? -K * u : K * u; }

void orc_fold_table(double *cK, int n_steps, double c0, double g)
{
    double c = c0;
    cK[0] = c;
    for (int t = 1; t <= n_steps; ++t) { c *= g; cK[t] = c; }
}

/* the two constants from the float32 drift exponent / start value of the generator (orc_gbm_paths_f32 above) */
void orc_fold_constants(double S0, double K, double r, double sigma, double T, int n_steps, double *c0, double *g)
{
    const double dt = T / n_steps, L2E = 1.4426950408889634074;
    const float a = (float)((r - 0.5 * sigma * sigma) * dt * L2E);
    const float s0 = (float)S0;
    *c0 = (double)s0 * (double)s0 / K;
    *g = exp2(2.0 * (double)a);
}

int orc_lsm_two_pass_folded(const float *S, int64_t ld, int64_t n_pairs, int n_steps, double K, double r, double T,
                            int is_put, double c0, double g, orc_lsm_result *res, double *betas_out,
                            int64_t *nitm_out, int32_t *texa_out, int32_t *texb_out)
{
    const int N = n_steps;
    const int64_t P = n_pairs;
    const double dt = T / N, invK = 1.0 / K;
    double *D = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    double *cK = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    double *betas = (double *)calloc((size_t)(N + 1) * 3, sizeof(double));
    int64_t *nitm = (int64_t *)calloc((size_t)(N + 1), sizeof(int64_t));
    if (!D || !cK || !betas || !nitm) return -1;
    for (int k = 0; k <= N; ++k) D[k] = exp(-r * dt * (double)k);
    orc_fold_table(cK, N, c0, g);
    const float *SN = S + (int64_t)N * ld;
    for (int t = N - 1; t >= 1; --t) {
        const float *St = S + (int64_t)t * ld;
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
#pragma omp parallel for schedule(static) reduction(+ : m0, m1, m2, m3, m4, m5, m6, m7) if (P >= 131072)
        for (int64_t j = 0; j < P; ++j) {
            const double imm = payoff_d((double)St[j], K, is_put);
            if (imm > 0.0) {
                const double pN = payoff_d((double)SN[j], K, is_put);
                const double y = (pN > 0.0 ? pN : 0.0) * D[N - t];
                const double u = fma((double)St[j], invK, -1.0), u2 = u * u;
                m0 += 1.0; m1 += u; m2 += u2; m3 += u2 * u; m4 += u2 * u2;
                m5 += y; m6 += u * y; m7 += u2 * y;
            }
            const double ub = fma(cK[t], 1.0 / (double)St[j], -1.0);
            if (fold_pay(ub, K, is_put) > 0.0) {
                const double pN = fold_pay(fma(cK[N], 1.0 / (double)SN[j], -1.0), K, is_put);
                const double y = (pN > 0.0 ? pN : 0.0) * D[N - t];
                const double u2 = ub * ub;
                m0 += 1.0; m1 += ub; m2 += u2; m3 += u2 * ub; m4 += u2 * u2;
                m5 += y; m6 += ub * y; m7 += u2 * y;
            }
        }
        double m[8] = {m0, m1, m2, m3, m4, m5, m6, m7};
        nitm[t] = (int64_t)m[0];
        solve_poly2(m, betas + 3 * t);
    }
    double sum = 0.0, sumsq = 0.0;
    int64_t nex = 0, nzero = 0, snitm = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum, sumsq, nex, nzero) if (P >= 131072)
    for (int64_t j = 0; j < P; ++j) {
        int ta = N, tb = N;
        double pa = payoff_d((double)SN[j], K, is_put);
        double pb = fold_pay(fma(cK[N], 1.0 / (double)SN[j], -1.0), K, is_put);
        for (int t = N - 1; t >= 1 && (ta == N || tb == N); --t) {
            if (nitm[t] == 0) continue;
            const double *b = betas + 3 * t;
            const double st = (double)S[(int64_t)t * ld + j];
            if (ta == N) {
                const double imm = payoff_d(st, K, is_put);
                const double u = fma(st, invK, -1.0);
                if (imm > 0.0 && imm > fma(u, fma(u, b[2], b[1]), b[0])) { ta = t; pa = imm; }
            }
            if (tb == N) {
                const double ub = fma(cK[t], 1.0 / st, -1.0);
                const double imm = fold_pay(ub, K, is_put);
                if (imm > 0.0 && imm > fma(ub, fma(ub, b[2], b[1]), b[0])) { tb = t; pb = imm; }
            }
        }
        const double cfa = (pa > 0.0 ? pa : 0.0) * D[ta - 1], cfb = (pb > 0.0 ? pb : 0.0) * D[tb - 1];
        sum += cfa + cfb; sumsq += cfa * cfa + cfb * cfb;
        nex += (ta < N) + (tb < N);
        nzero += (cfa == 0.0) + (cfb == 0.0);
        if (texa_out) texa_out[j] = ta;
        if (texb_out) texb_out[j] = tb;
    }
    for (int t = 1; t < N; ++t) snitm += nitm[t];
    if (res) {
        res->sum = sum; res->sumsq = sumsq; res->price = sum / (double)(2 * P);
        res->n_paths = 2 * P; res->n_exercised = nex; res->n_zero = nzero; res->sum_nitm = snitm;
    }
    if (betas_out) memcpy(betas_out, betas, sizeof(double) * (size_t)(N + 1) * 3);
    if (nitm_out) memcpy(nitm_out, nitm, sizeof(int64_t) * (size_t)(N + 1));
    free(D); free(cK); free(betas); free(nitm);
    return 0;
}

/* European discounted payoff sums straight from a stored path matrix's last row */
void orc_european_from_paths(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K,
                             double r, double T, int is_put, double *sum, double *sumsq)
{
    const double df = exp(-r * T);
    double s = 0.0, q = 0.0;
    for (int64_t j = 0; j < n_paths; ++j) {
        double p = payoff_d((double)S[(int64_t)n_steps * ld + j], K, is_put);
        double cf = (p > 0.0 ? p : 0.0) * df;
        s += cf; q += cf * cf;
    }
    *sum = s; *sumsq = q;
}

/* ---- pieces of the two-pass flow exposed separately (multi-process sharding tests):
 * pass-1 moments [n_steps+1][8] of a shard, and the per-step solve on (all-reduced) moments */
void orc_lsm_pass1_moments(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K,
                           double r, double T, int is_put, double *moments)
{
    const int N = n_steps;
    const double dt = T / N, invK = 1.0 / K;
    const float *SN = S + (int64_t)N * ld;
    memset(moments, 0, sizeof(double) * 8 * (size_t)(N + 1));
    for (int t = N - 1; t >= 1; --t) {
        const float *St = S + (int64_t)t * ld;
        double *m = moments + 8 * (size_t)t;
        const double d = exp(-r * dt * (double)(N - t));
        for (int64_t j = 0; j < n_paths; ++j) {
            double imm = payoff_d((double)St[j], K, is_put);
            if (!(imm > 0.0)) continue;
            double pN = payoff_d((double)SN[j], K, is_put);
            double y = (pN > 0.0 ? pN : 0.0) * d;
            double u = fma((double)St[j], invK, -1.0), u2 = u * u;
            m[0] += 1.0; m[1] += u; m[2] += u2; m[3] += u2 * u; m[4] += u2 * u2;
            m[5] += y; m[6] += u * y; m[7] += u2 * y;
        }
    }
}

void orc_solve_poly2(const double m[8], double beta[3]) { solve_poly2(m, beta); }


/* terminal spots only (the calibrator's inner simulation), same Philox layout as the paths */
void orc_heston_terminal_f32(float *ST, int64_t n_paths, int n_steps, double S0, double r, double T,
                             double v0, double kappa, double theta, double xi, double rho,
                             uint64_t seed, uint32_t stream, uint64_t pair_offset, int scheme)
{
    const heston_consts c = heston_make(r, T, n_steps, kappa, theta, xi, rho);
    const int64_t P = n_paths / 2;
#pragma omp parallel for schedule(static) if (P * (int64_t)n_steps >= 1000000)
    for (int64_t p = 0; p < P; ++p) {
        float s0 = (float)S0, s1 = (float)S0, va = (float)v0, vb = (float)v0, z[4];
        for (int t = 0; t < n_steps; ++t) {
            int i = t & 1;
            if (i == 0) orc_normals4(seed, pair_offset + (uint64_t)p, (uint32_t)(t >> 1), stream, z);
            heston_step(&c, scheme, z[2 * i], z[2 * i + 1], &s0, &va);
            heston_step(&c, scheme, -z[2 * i], -z[2 * i + 1], &s1, &vb);
        }
        ST[p] = s0;
        ST[p + P] = s1;
    }
}
