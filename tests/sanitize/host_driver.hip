// Host-side sanitizer driver (tests/test_sanitize_cpu.py builds it with -fsanitize=address,undefined for
// the HOST only and runs it without a GPU): the C ABI's argument checks and error paths, the batched
// path's table / slab planning and the per-step sweep's argument image -- everything in libomc.so that
// runs on the CPU and can be reached without a device.  GPU sanitizers are not available on the pool;
// this is the CPU build SURVEY.md section 5.2 asks for.  Prints "host_driver ok" and exits 0.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/omc.h"
#include "../../options_model_amd/csrc/omc_batch.h"
#include "../../options_model_amd/csrc/omc_kernels.h"

#define REQUIRE(cond)                                                                 \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "REQUIRE failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

static omc_params make(int model, int sem, int64_t M, int N, double T = 1.0)
{
    omc_params p;
    memset(&p, 0, sizeof p);
    p.model = model; p.is_put = 1; p.semantics = sem; p.antithetic = 1; p.heston_scheme = 0;
    p.n_steps = N; p.n_paths = M;
    p.S0 = 100; p.K = 100; p.r = 0.05; p.sigma = 0.2; p.T = T;
    p.v0 = 0.04; p.kappa = 2; p.theta = 0.04; p.xi = 0.3; p.rho = -0.7;
    p.seed = 42;
    return p;
}

int main()
{
    // ---- C ABI without a device: every call must fail cleanly, never crash
    REQUIRE(omc_abi_version() == OMC_ABI_VERSION);
    int n = -1;
    (void)omc_device_count(&n);
    REQUIRE(n >= 0);
    REQUIRE(omc_device_count(nullptr) != 0);
    omc_ctx* ctx = (omc_ctx*)0x1;
    const int rc = omc_ctx_create(0, nullptr, &ctx);
    if (n == 0) {
        REQUIRE(rc != 0 && ctx == nullptr);  // the failure branch releases what it had created
        REQUIRE(strlen(omc_last_error()) > 0);
    }
    REQUIRE(omc_ctx_create(0, nullptr, nullptr) != 0);
    REQUIRE(omc_ctx_destroy(nullptr) == 0);
    omc_result res;
    omc_params p = make(0, 2, 1000, 10);
    REQUIRE(omc_price_american(nullptr, &p, &res, nullptr, 0) != 0);
    REQUIRE(omc_price_american_seq(nullptr, &p, 1, &res) != 0);
    REQUIRE(omc_price_american_batch(nullptr, &p, 1, &res) != 0);
    REQUIRE(omc_price_american_contnet(nullptr, &p, 32, 10, 1e-3, 1, &res) != 0);
    REQUIRE(omc_lsm_contnet(nullptr, nullptr, 0, 1000, 10, 100.0, 0.05, 1.0, 1, 32, 10, 1e-3, 1, &res, nullptr, nullptr) != 0);
    float net[8];
    REQUIRE(omc_contnet_init_params(nullptr, 32, 1, 1, net, 8) != 0);
    REQUIRE(omc_set_option(nullptr, "gbm_vec", 1) != 0);
    REQUIRE(omc_set_allreduce_hook(nullptr, nullptr, nullptr) != 0);
    REQUIRE(omc_comm_info(nullptr, nullptr, nullptr) != 0);
    char small[8];
    REQUIRE(omc_comm_unique_id(small, sizeof small) == -7);
    REQUIRE(omc_comm_unique_id(nullptr, 128) == -7);
    REQUIRE(omc_comm_init(nullptr, 0, 1, small, sizeof small) != 0);
    REQUIRE(omc_mlp_param_count(64, 2) > 0 && omc_mlp_param_count(63, 2) < 0);
    REQUIRE(omc_localvol_param_count(64, 4) > 0);
    REQUIRE(omc_mlp_train_supported(128, 3, 256) == 1 && omc_mlp_train_supported(32, 2, 256) == 1 && omc_mlp_train_supported(32, 3, 256) == 1 && omc_mlp_train_supported(48, 2, 256) == 0 && omc_mlp_train_supported(256, 3, 256) == 0);
    // round 5's entry points: which trainer kernel a minibatch runs (host arithmetic), and the failure branches of the rest
    REQUIRE(omc_mlp_train_variant(128, 3, 256) == 4 && omc_mlp_train_variant(128, 3, 4096) == 4 && omc_mlp_train_variant(128, 3, 4097) == 3);
    REQUIRE(omc_mlp_train_variant(128, 3, 8193) == 2 && omc_mlp_train_variant(64, 2, 1024) == 4 && omc_mlp_train_variant(64, 2, 1025) == 1);
    REQUIRE(omc_mlp_train_variant(32, 2, 5000) == 3 && omc_mlp_train_variant(48, 2, 256) == 0 && omc_mlp_train_variant(64, 4, 256) == 0);
    unsigned char mask[64];
    REQUIRE(omc_mlp_dropout_masks(nullptr, 4, 64, 1, 1, nullptr, 1, 1, 0.1, mask) != 0);
    REQUIRE(omc_ctx_device_info(nullptr, nullptr, nullptr, 0, nullptr, 0) != 0);
    REQUIRE(omc_lsm_ols7(nullptr, nullptr, 0, 1000, 10, 100.0, 0.05, 1.0, 1, &res, nullptr, nullptr, nullptr, nullptr) != 0);
    REQUIRE(omc_price_american_ols7(nullptr, &p, &res, nullptr, nullptr) != 0);
    if (ctx) omc_ctx_destroy(ctx);

    // ---- batched path: slab planning and the per-problem table (host arithmetic only; the "device"
    // pointers are offsets into a host slab that is never dereferenced as such)
    std::vector<omc_params> items;
    const int64_t sizes[] = {2, 10, 1000, 1002, 10000, 65536, 100000};
    const int steps[] = {1, 2, 10, 50, 130};
    for (int64_t M : sizes)
        for (int N : steps) items.push_back(make(0, 0, M, N, 0.01 * N));
    for (int american = 0; american < 2; ++american)
        for (int two_pass = 0; two_pass < 2; ++two_pass) {
            const int nitems = (int)items.size();
            for (auto& it : items) it.semantics = two_pass ? 2 : 0;
            const size_t slab = omc::batch_slab_bytes(items.data(), nitems, american, two_pass);
            const size_t tab = omc::batch_table_bytes(nitems);
            const size_t nd = omc::batch_discount_doubles(items.data(), nitems);
            REQUIRE(slab > 0 && tab > 0 && nd > 0);
            std::vector<char> table(tab);            // exact sizes: an overrun is an ASan report
            std::vector<double> disc(nd), results(8 * (size_t)nitems);
            std::vector<char> fake_slab(16);         // base address only
            omc::BatchExtents e;
            omc::batch_build(items.data(), nitems, american, two_pass, fake_slab.data(), results.data(), disc.data(),
                             table.data(), disc.data(), &e);
            if (american) {
                REQUIRE(e.max_steps == 130 && e.path_blocks > 0 && e.sweep_blocks > 0 && e.block_blocks > 0);
                for (size_t k = 0; k < nd; ++k) REQUIRE(disc[k] > 0.0 && disc[k] <= 1.0);
            } else {
                REQUIRE(e.term_blocks > 0);
            }
        }
    // a batch that mixes vector-width-friendly and odd sizes falls back to scalar accesses
    {
        std::vector<omc_params> mix = {make(0, 2, 4096, 20), make(0, 2, 1002, 20)};
        omc::BatchExtents e;
        std::vector<char> table(omc::batch_table_bytes(2)), slab0(16);
        std::vector<double> disc(omc::batch_discount_doubles(mix.data(), 2)), results(16);
        omc::batch_build(mix.data(), 2, true, true, slab0.data(), results.data(), disc.data(), table.data(), disc.data(), &e);
        REQUIRE(e.vec4 == 0);
    }

    // ---- per-step sweep: geometry helpers and the device argument image
    // per-step network flow: scratch carving and the trainer width that hosts a net
    REQUIRE(omc::cn_padded_width(1) == 32 && omc::cn_padded_width(32) == 32 && omc::cn_padded_width(33) == 64 &&
            omc::cn_padded_width(128) == 128 && omc::cn_padded_width(129) < 0);
    for (int64_t M : {int64_t(1), int64_t(2047), int64_t(2048), int64_t(2049), int64_t(8000000)}) {
        omc::LsmProblem lp{nullptr, M, M, 10, 1, 100.0, 0.05, 1.0};
        std::vector<char> scratch(omc::cn_scratch_bytes(M));
        const char* hdr = (const char*)omc::cn_header(lp, scratch.data());
        REQUIRE(hdr >= scratch.data() && hdr + 32 <= scratch.data() + scratch.size());
        REQUIRE(((uintptr_t)(hdr - scratch.data()) & 7) == 0);
    }
    REQUIRE(omc::mlp_partial_bytes(32, 2, 32 * 5000) == sizeof(float) * 5000 * 1408);
    REQUIRE(omc::lsm_sweep_blocks(1) == 1);
    REQUIRE(omc::lsm_sweep_blocks(1000000) <= 256 && omc::lsm_sweep_blocks(64000000) <= 256);
    REQUIRE(omc::lsm_step_blocks(1000000) <= omc::kMaxLsmBlocks);
    REQUIRE(omc::lsm_part1_tiles(1000000) >= 977);
    {
        std::vector<char> img(omc::lsm_sweep_args_bytes()), img2(omc::lsm_sweep_args_bytes());
        omc::LsmProblem lp{(const float*)0x1000, 1024, 1000, 50, 1, 100.0, 0.05, 1.0};
        omc::LsmWorkspace w;
        memset((void*)&w, 0, sizeof w);
        omc::lsm_sweep_args_image(lp, w, 0, false, img.data());
        omc::lsm_sweep_args_image(lp, w, 0, false, img2.data());
        REQUIRE(img == img2);  // no uninitialised padding leaks into the compared image
        omc::lsm_sweep_args_image(lp, w, 0, true, img2.data());
        REQUIRE(img != img2);
    }
    printf("host_driver ok\n");
    return 0;
}
