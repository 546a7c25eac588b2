/* Prints the layout of the C ABI's structs as the C compiler sees them (JSON); the test compares it with
 * the ctypes mirrors in options_model_amd/_ffi.py. */
#include <stddef.h>
#include <stdio.h>

#include "../../include/omc.h"

#define F(type, field) printf("  \"%s.%s\": [%zu, %zu],\n", #type, #field, offsetof(type, field), sizeof(((type*)0)->field))

int main(void)
{
    printf("{\n");
    F(omc_params, model); F(omc_params, is_put); F(omc_params, semantics); F(omc_params, antithetic);
    F(omc_params, heston_scheme); F(omc_params, n_steps); F(omc_params, n_paths);
    F(omc_params, S0); F(omc_params, K); F(omc_params, r); F(omc_params, sigma); F(omc_params, T);
    F(omc_params, v0); F(omc_params, kappa); F(omc_params, theta); F(omc_params, xi); F(omc_params, rho);
    F(omc_params, seed); F(omc_params, stream); F(omc_params, pair_offset);
    F(omc_result, price); F(omc_result, sum); F(omc_result, sumsq); F(omc_result, std); F(omc_result, zero_prob);
    F(omc_result, n_paths); F(omc_result, n_exercised); F(omc_result, n_zero); F(omc_result, sum_nitm);
    F(omc_result, ms_paths); F(omc_result, ms_lsm); F(omc_result, ms_total); F(omc_result, ms_pass1);
    F(omc_result, ms_pass2); F(omc_result, timed); F(omc_result, folded);
    printf("  \"sizeof.omc_params\": [%zu, 0],\n  \"sizeof.omc_result\": [%zu, 0],\n", sizeof(omc_params), sizeof(omc_result));
    printf("  \"abi\": [%d, 0]\n}\n", OMC_ABI_VERSION);
    return 0;
}
