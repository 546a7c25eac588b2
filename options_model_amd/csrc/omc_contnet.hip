// omc_contnet.hip -- the regressor the reference's v1 / v2 pricers actually run at every time step
// (Options_model.py:112-151, options_model_2.py:283-312): a FRESH ContNet(1 -> h -> h -> 1) per step,
// input = the spots of the step's regression set standardised by their own mean / population std
// (std 0: only centred), target = the set's cash-flows valued at the step (NOT normalised), `epochs`
// full-batch Adam steps on the mean squared error, then continuation = net(input) and the strict
// `payoff > continuation` decision.
//
// What this file adds per step t, around pieces that already exist:
//   cn_count_kernel   members of the set (in the money at t, not yet exercised), sums of (S - K), (S - K)^2
//   cn_scan_kernel    row offsets per workgroup; n, mean, 1/std  (one small workgroup)
//   cn_rows_kernel    the set as training rows [xs, 0 x 6, y] in path order (the trainer's row format)
//   cn_init_kernel    PyTorch's default nn.Linear initialisation (U(-1/sqrt(fan_in), +1/sqrt(fan_in)) for
//                     weights and biases) from Philox keyed by (seed, t); Adam moments zeroed
//   [omc_mlp.hip]     `epochs` x mlp_train_steps with batch = n: ContNet embeds EXACTLY in the trainer's
//                     8-input layout (input 0 = xs, inputs 1..6 = 0 with zero weights, input 7 = the bias
//                     column) and in its 32-unit granularity (units >= h have zero weights, so they output
//                     0, receive zero gradients and stay zero under Adam)
//   cn_forward_kernel continuation value of every member -> one row of float32
//   [omc_lsm.hip]     lsm_step in "values" mode: the decision, mask, state -- the code pinned by the
//                     reference's recorded per-step runs
// Row order, partial sums and the Adam reduction are fixed, so a pricing is bitwise reproducible.
#include "omc_contnet_dev.h"

namespace omc {

namespace {

__global__ __launch_bounds__(kCnBlock) void cn_count_kernel(CnArgs a) { cn_count_body(a); }
__global__ __launch_bounds__(1024) void cn_scan_kernel(CnArgs a) { cn_scan_body(a); }
__global__ __launch_bounds__(kCnBlock) void cn_rows_kernel(CnArgs a) { cn_rows_body(a); }
__global__ __launch_bounds__(256) void cn_init_kernel(CnInitArgs a) { cn_init_one(a, blockIdx.x * blockDim.x + threadIdx.x); }
template <int H>
__global__ __launch_bounds__(kCnBlock) void cn_forward_kernel(CnFwdArgs fa) { cn_forward_body<H>(fa); }

}  // namespace

int cn_blocks(int64_t M) { return (int)((M + kCnSpan - 1) / kCnSpan); }

size_t cn_scratch_bytes(int64_t M)
{
    const size_t nb = (size_t)cn_blocks(M);
    // cnt | s1 | s2 | offs | hdr
    return 8 * ((nb + 1) / 2 + 1) + 8 * nb + 8 * nb + 8 * (nb + 1) + 8 * 4;
}

int cn_padded_width(int hidden) { return hidden <= 32 ? 32 : hidden <= 64 ? 64 : hidden <= 128 ? 128 : -1; }

static void carve(const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt, float* data, float* cont,
                  CnArgs* a)
{
    const size_t nb = (size_t)cn_blocks(p.M);
    char* q = (char*)scratch;
    a->cnt = (int32_t*)q; q += 8 * ((nb + 1) / 2 + 1);
    a->s1 = (double*)q; q += 8 * nb;
    a->s2 = (double*)q; q += 8 * nb;
    a->offs = (int64_t*)q; q += 8 * (nb + 1);
    a->hdr = (double*)q;
    a->St = p.S + (size_t)t * p.ld;
    a->SN = p.S + (size_t)p.N * p.ld;
    a->live = w.live;
    a->M = p.M;
    a->K = p.K;
    a->Dt = Dt;
    a->is_put = p.is_put;
    a->nblk = (int)nb;
    a->data = data;
    a->cont = cont;
}

const double* cn_header(const LsmProblem& p, void* scratch)
{
    const size_t nb = (size_t)cn_blocks(p.M);
    return (const double*)((char*)scratch + 8 * ((nb + 1) / 2 + 1) + 8 * nb + 8 * nb + 8 * (nb + 1));
}

hipError_t cn_count(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt)
{
    CnArgs a;
    carve(p, w, scratch, t, Dt, nullptr, nullptr, &a);
    hipLaunchKernelGGL(cn_count_kernel, dim3(a.nblk), dim3(kCnBlock), 0, st, a);
    hipLaunchKernelGGL(cn_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t cn_rows(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt,
                   float* data)
{
    CnArgs a;
    carve(p, w, scratch, t, Dt, data, nullptr, &a);
    hipLaunchKernelGGL(cn_rows_kernel, dim3(a.nblk), dim3(kCnBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t cn_init(hipStream_t st, int hidden, int t, uint64_t seed, float* params, float* m, float* v)
{
    CnInitArgs a;
    a.params = params; a.m = m; a.v = v;
    a.H = cn_padded_width(hidden);
    if (a.H < 0) return hipErrorInvalidValue;
    a.h = hidden;
    a.np = a.H * 8 + a.H * a.H + a.H + a.H + 1;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
    a.t = (uint32_t)t;
    hipLaunchKernelGGL(cn_init_kernel, dim3((a.np + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

template <int H>
static hipError_t forward_launch(hipStream_t st, const CnFwdArgs& fa)
{
    constexpr size_t lds = sizeof(float) * (size_t)(H * 8 + H * H + H + H + 1);
    static std::atomic<uint64_t> attr_mask{0};  // per device (omc_kernels.h)
    if (lds > 48 * 1024) {
        hipError_t e = set_max_dynamic_lds(attr_mask, reinterpret_cast<const void*>(cn_forward_kernel<H>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((cn_forward_kernel<H>), dim3((unsigned)((fa.c.M + kCnBlock - 1) / kCnBlock)), dim3(kCnBlock), lds,
                       st, fa);
    return hipGetLastError();
}

hipError_t cn_forward(hipStream_t st, const LsmProblem& p, const LsmWorkspace& w, void* scratch, int t, double Dt,
                      int hidden, const float* params, float* cont)
{
    CnFwdArgs fa;
    carve(p, w, scratch, t, Dt, nullptr, cont, &fa.c);
    fa.params = params;
    fa.h = hidden;
    switch (cn_padded_width(hidden)) {
    case 32: return forward_launch<32>(st, fa);
    case 64: return forward_launch<64>(st, fa);
    case 128: return forward_launch<128>(st, fa);
    }
    return hipErrorInvalidValue;
}

}  // namespace omc
