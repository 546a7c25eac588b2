/* Minimal C host for libomc.so (include/omc.h): price the BASELINE config-2 option on GPU 0.
 *
 *   gcc -O2 -I include examples/price_american.c -o /tmp/price_american \
 *       -L options_model_amd/lib -lomc -lm -Wl,-rpath,$PWD/options_model_amd/lib
 *   /tmp/price_american [n_paths] [n_steps]
 *
 * Nothing here is Python- or PyTorch-specific: this is the boundary a host in any language binds. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "omc.h"

int main(int argc, char** argv)
{
    omc_ctx* ctx = NULL;
    int rc = omc_ctx_create(0, NULL, &ctx);
    if (rc != 0) {
        fprintf(stderr, "omc_ctx_create: %d (%s)\n", rc, omc_last_error());
        return 1;
    }
    omc_params p;
    memset(&p, 0, sizeof p);
    p.model = OMC_MODEL_GBM;
    p.is_put = 1;
    p.semantics = OMC_SEM_TWO_PASS;
    p.antithetic = 1;
    p.n_paths = argc > 1 ? atoll(argv[1]) : 1000000;
    p.n_steps = argc > 2 ? atoi(argv[2]) : 252;
    p.S0 = 100.0; p.K = 100.0; p.r = 0.05; p.sigma = 0.2; p.T = 1.0;
    p.seed = 42;
    omc_result res;
    rc = omc_price_american(ctx, &p, &res, NULL, 0);
    if (rc != 0) {
        fprintf(stderr, "omc_price_american: %d (%s)\n", rc, omc_last_error());
        omc_ctx_destroy(ctx);
        return 1;
    }
    printf("price %.6f  stderr %.6f  paths %lld  exercised %lld  kernels: paths %.3f ms, lsm %.3f ms  storage: %s\n", res.price,
           res.std / sqrt((double)(res.n_paths > 0 ? res.n_paths : 1)), (long long)res.n_paths,
           (long long)res.n_exercised, res.ms_paths, res.ms_lsm, res.folded ? "antithetic-folded" : "full");
    omc_ctx_destroy(ctx);
    return 0;
}
