/* Sanitizer driver for the CPU oracle (tests/test_sanitize_cpu.py builds oracle/omc_oracle.c together
 * with this file under -fsanitize=address,undefined and runs it): every entry point on exactly-sized
 * heap buffers, including the smallest shapes (2 paths, 1 step), odd sizes and strides. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double price, sum, sumsq;
    int64_t n_paths, n_exercised, n_zero, sum_nitm;
} orc_lsm_result;

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_gbm_normals_f32(float *Z, int64_t ldz, int64_t n_pairs, int n_steps, uint64_t seed, uint32_t stream,
                         uint64_t pair_offset);
void orc_gbm_paths_f32(float *S, int64_t ld, int64_t n_paths, int n_steps, double S0, double r, double sigma,
                       double T, uint64_t seed, uint32_t stream, uint64_t pair_offset, int antithetic);
void orc_gbm_paths_from_normals_f32(float *S, int64_t ld, int64_t n_paths, int n_steps, double S0, double r,
                                    double sigma, double T, const float *Zhalf, int64_t ldz, int antithetic);
void orc_heston_paths_f32(float *S, int64_t ld, int64_t n_paths, int n_steps, double S0, double r, double T,
                          double v0, double kappa, double theta, double xi, double rho, uint64_t seed,
                          uint32_t stream, uint64_t pair_offset, int scheme);
void orc_heston_terminal_f32(float *ST, int64_t n_paths, int n_steps, double S0, double r, double T, double v0,
                             double kappa, double theta, double xi, double rho, uint64_t seed, uint32_t stream,
                             uint64_t pair_offset, int scheme);
int orc_lsm_poly(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K, double r, double T,
                 int is_put, int semantics, orc_lsm_result *res, double *betas_out, int64_t *nitm_out,
                 float *sx_out, int32_t *tex_out);
int orc_lsm_apply_frozen(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K, double r, double T,
                         int is_put, const double *betas, const int64_t *nitm, orc_lsm_result *res,
                         float *sx_out, int32_t *tex_out);
void orc_lsm_pass1_moments(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K, double r,
                           double T, int is_put, double *m);
void orc_european_from_paths(const float *S, int64_t ld, int64_t n_paths, int n_steps, double K, double r,
                             double T, int is_put, double *sum, double *sumsq);

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "REQUIRE failed: %s (line %d)\n", #c, __LINE__); exit(1); } } while (0)

int main(void)
{
    uint32_t ctr[4] = {0, 0, 0, 0}, key[2] = {0, 0}, out[4];
    orc_philox4x32_10(ctr, key, out);
    REQUIRE(out[0] == 0x6627e8d5u && out[3] == 0x9b00dbd8u);

    const int64_t sizes[] = {2, 6, 10, 1000, 1002, 4096};
    const int steps[] = {1, 2, 7, 50};
    for (unsigned a = 0; a < sizeof sizes / sizeof *sizes; ++a)
        for (unsigned b = 0; b < sizeof steps / sizeof *steps; ++b) {
            const int64_t M = sizes[a];
            const int N = steps[b];
            float *S = malloc(sizeof(float) * (size_t)M * (size_t)(N + 1));
            float *Z = malloc(sizeof(float) * (size_t)(M / 2) * (size_t)N);
            orc_gbm_normals_f32(Z, M / 2, M / 2, N, 42, 1, 3);
            orc_gbm_paths_from_normals_f32(S, M, M, N, 100, 0.05, 0.2, 1.0, Z, M / 2, 1);
            orc_gbm_paths_f32(S, M, M, N, 100, 0.05, 0.2, 1.0, 42, 0, 5, 0);  /* non-antithetic */
            orc_gbm_paths_f32(S, M, M, N, 100, 0.05, 0.2, 1.0, 42, 0, 5, 1);
            double *betas = malloc(sizeof(double) * 3 * (size_t)(N + 1));
            int64_t *nitm = malloc(sizeof(int64_t) * (size_t)(N + 1));
            float *sx = malloc(sizeof(float) * (size_t)M);
            int32_t *tex = malloc(sizeof(int32_t) * (size_t)M);
            double *mom = malloc(sizeof(double) * 8 * (size_t)(N + 1));
            for (int put = 0; put < 2; ++put)
                for (int sem = 0; sem < 3; ++sem) {
                    orc_lsm_result r;
                    REQUIRE(orc_lsm_poly(S, M, M, N, 100, 0.05, 1.0, put, sem, &r, betas, nitm, sx, tex) == 0);
                    REQUIRE(r.price >= 0 && isfinite(r.price) && r.n_paths == M);
                    REQUIRE(orc_lsm_poly(S, M, M, N, 100, 0.05, 1.0, put, sem, &r, NULL, NULL, NULL, NULL) == 0);
                    if (sem == 2) {
                        orc_lsm_result q;
                        REQUIRE(orc_lsm_apply_frozen(S, M, M, N, 100, 0.05, 1.0, put, betas, nitm, &q, sx, tex) == 0);
                        REQUIRE(q.n_exercised == r.n_exercised);
                    }
                }
            orc_lsm_pass1_moments(S, M, M, N, 100, 0.05, 1.0, 1, mom);
            double s1, s2;
            orc_european_from_paths(S, M, M, N, 100, 0.05, 1.0, 1, &s1, &s2);
            REQUIRE(s1 >= 0 && s2 >= 0);
            for (int scheme = 0; scheme < 3; ++scheme) {
                orc_heston_paths_f32(S, M, M, N, 100, 0.05, 1.0, 0.04, 2, 0.04, scheme == 1 ? 1.0 : 0.3, -0.7, 7, 2, 1, scheme);
                float *ST = malloc(sizeof(float) * (size_t)M);
                orc_heston_terminal_f32(ST, M, N, 100, 0.05, 1.0, 0.04, 2, 0.04, 0.3, -0.7, 7, 2, 1, scheme);
                if (scheme < 2) REQUIRE(ST[0] > 0);
                free(ST);
            }
            free(S); free(Z); free(betas); free(nitm); free(sx); free(tex); free(mom);
        }
    printf("oracle_driver ok\n");
    return 0;
}
