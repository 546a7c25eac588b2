// omc_batch.h -- host interface of the batched (many small problems) launch path.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/omc.h"

namespace omc {

using BatchItem = omc_params;  // one problem = one omc_params

struct BatchExtents {  // launch extents = maxima over the problems of a batch
    int max_steps;
    int64_t path_blocks;  // 256-thread blocks at one pair per thread
    int sweep_blocks;     // per-step sweep (512-thread blocks)
    int block_blocks;     // pass 2 / valuation (256-thread blocks)
    int64_t tile_blocks;  // pass 1 tile groups / 4
    int term_blocks;      // terminal-only kernel
    int vec4;             // every problem admits 16-byte accesses
};

size_t batch_slab_bytes(const BatchItem* items, int n, bool american, bool two_pass);
size_t batch_table_bytes(int n);
size_t batch_discount_doubles(const BatchItem* items, int n);
void batch_build(const BatchItem* items, int n, bool american, bool two_pass, char* slab,
                 double* results_dev, double* disc_dev, void* table_host, double* disc_host,
                 BatchExtents* ext);
// gen: 0 GBM antithetic, 1 GBM plain, 2 Heston reference clamp, 3 Heston full truncation, 4 Heston calibrator scheme
hipError_t batch_paths(hipStream_t st, const void* table_dev, int n, const BatchExtents& e, int gen);
hipError_t batch_lsm(hipStream_t st, const void* table_dev, int n, const BatchExtents& e, int semantics);
hipError_t batch_terminal(hipStream_t st, const void* table_dev, int n, const BatchExtents& e, int gen);

// ---- the per-step ContNet flow (the v1 / v2 pricers' regressor) for a whole batch: second slab + two more tables
struct MlpBatchJob;
size_t batch_cn_table_bytes(int n);
size_t batch_cn_slab_bytes(const BatchItem* items, int n, int hidden);
void batch_cn_build(const BatchItem* items, int n, int hidden, const uint64_t* seeds, double lr, char* slab2,
                    void* table_host, void* cn_table_host, MlpBatchJob* jobs, int* max_cn_blocks, int64_t* max_paths);
hipError_t batch_contnet(hipStream_t st, const void* table_dev, const void* cn_table_dev, const void* mlp_table_dev,
                         int n, const BatchExtents& e, int hidden, int epochs, int max_cn_blocks, int64_t max_paths,
                         const double* bc1_dev, const double* bc2_dev, int* tile_prefix_dev /* n + 1 ints */);

}  // namespace omc
