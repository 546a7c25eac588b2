// omc_mlp.hip -- the continuation-value network of the NN flow as
// hand-written gfx950 kernels: one fused forward + backward pass of the 7 -> 64 -> 64 (-> 64) -> 1
// MLP (Linear/ReLU/Dropout stacks of SingleLSMNet, options_model_3.py:85-103; the reference's class
// always has three hidden layers, BASELINE config 5 names two) over a minibatch, and
// the gradient reduce + Adam update (options_model_3.py:565-600: Adam, weight decay, MSE).
//
// The minibatch GEMMs run on the matrix cores in float32 (v_mfma_f32_32x32x2_f32; the reference
// trains in float32).  Everything is kept TRANSPOSED -- activations are [hidden][row] -- so that
// the accumulator layout of one layer (lane <-> batch row, registers <-> hidden units) is
// directly the B operand of the next layer's MFMA: the k-index of a matrix product may be
// walked in any order as long as A and B agree, and the order chosen here is the one the
// accumulator registers already have.  Only the weight-gradient products, which contract over
// the batch rows, need the operands turned around; they go through a wave-private LDS patch.
//
// One wave owns a tile of 32 batch rows from load to weight gradients; a workgroup (4 waves,
// one per SIMD, up to 512 registers each) shares the weights in LDS.  Gradients accumulate in
// registers over all tiles of a wave, are summed over the workgroup in a fixed order and
// written as one partial per workgroup; the Adam kernel sums the partials in index order.
// No atomics: a training run is bitwise reproducible.
#include "omc_device.h"
#include "omc_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace omc {

namespace {

constexpr int kH = 64;                 // hidden width
constexpr int kLdW1 = 9, kLdW2 = 65;   // LDS leading dimensions (odd: conflict-free column walks)
// Staging patches are [unit][32 rows] without padding; the row index is XOR-swizzled per unit in
// units of 4 rows, so that the 16-byte row-quad reads of 16 consecutive units fall on 16 different
// bank groups and a quad stays contiguous.  (The last hidden layer's patch holds H, not dZ: the
// reading lane rebuilds dZ = [H > 0] wo dout / keep from it and gets the output-weight gradient
// from the same values, which saves the 32 per-lane accumulators a register-side sum would need.)
__device__ __forceinline__ int st_idx(int unit, int row) { return unit * 32 + (row ^ (((unit >> 1) & 7) << 2)); }
// The two access patterns of the kernel, written so that every address is one of four per-lane
// registers plus a compile-time offset (an XOR with a lane-dependent value cannot be folded into
// the instruction's immediate offset; left to itself hipcc keeps ~100 separate addresses live):
//   write: unit = unit_of(mt, r, h), row = c        -> 32*unit_const(mt, r) + wr[((r >> 2) & 1) * 2 + ((r >> 1) & 1)]
//   read : unit = c (+32), rows 16h + 4q .. + 3     -> 32*32*(unit >= 32) + rd[q]
struct StageOfs { int wr[4], rd[4]; };
__device__ __forceinline__ StageOfs stage_offsets(int c, int h)
{
    StageOfs o;
    // swizzle of unit_of(mt, r, h): ((unit >> 1) & 7) = 4*((r >> 2) & 1) ^ 2*h ^ ((r >> 1) & 1)
#pragma unroll
    for (int v = 0; v < 4; ++v) o.wr[v] = 128 * h + (c ^ (((4 * (v >> 1)) ^ (2 * h) ^ (v & 1)) << 2));
#pragma unroll
    for (int q = 0; q < 4; ++q) o.rd[q] = c * 32 + ((16 * h + 4 * q) ^ (((c >> 1) & 7) << 2));
    return o;
}
// offset of unit_of(mt, r, h) without its h term (that one is in StageOfs::wr)
__device__ __forceinline__ constexpr int unit_base(int mt, int r) { return 32 * (32 * mt + (r >> 2) * 8 + (r & 3)); }
__device__ __forceinline__ constexpr int wr_sel(int r) { return ((r >> 2) & 1) * 2 + ((r >> 1) & 1); }

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// hidden unit held by accumulator register r of 32x32 tile mt in half-wave h
__device__ __forceinline__ int unit_of(int mt, int r, int h) { return 32 * mt + (r >> 2) * 8 + 4 * h + (r & 3); }

__device__ __forceinline__ v16f mfma(float a, float b, v16f c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ReLU + inverted dropout on one layer's pre-activations (32 per lane), in place.
// One Philox block per (row, half-wave, layer, step) seeds two multiply-with-carry streams
// (x <- a * lo32(x) + hi32(x): one v_mad_u64_u32 per 32 bits); 16 bits per unit, kept if
// below keep16.
// SCALE = false leaves the 1 / keep factor to the caller (the trainer folds it into the next
// layer's weights).
template <bool SCALE>
__device__ __forceinline__ void relu_dropout(v16f (&z)[2], uint32_t row, uint32_t step, uint32_t tag,
                                             uint32_t keep16, float inv_keep, uint32_t k0, uint32_t k1)
{
    if (keep16 >= 65536u) {  // no dropout (uniform branch)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[mt][r] = fmaxf(z[mt][r], 0.0f);
        return;
    }
    const U4 o = philox4x32_10(row, step, tag, 0x4d4c5031u, k0, k1);
    uint64_t st[2] = {((uint64_t)o.x << 32) | (o.y | 1u), ((uint64_t)o.z << 32) | (o.w | 1u)};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            st[mt] = (uint64_t)4294957665u * (uint32_t)st[mt] + (st[mt] >> 32);
            const uint32_t w = (uint32_t)st[mt];
            const float v0 = z[mt][e], v1 = z[mt][e + 1];
            z[mt][e] = (v0 > 0.0f && (w & 0xffffu) < keep16) ? (SCALE ? v0 * inv_keep : v0) : 0.0f;
            z[mt][e + 1] = (v1 > 0.0f && (w >> 16) < keep16) ? (SCALE ? v1 * inv_keep : v1) : 0.0f;
        }
    }
}

// Pseudo-random permutation of [0, n): a 4-round Feistel network on ceil(log2 n) bits (the two
// halves may differ by one bit; an even number of rounds restores their order) with
// cycle-walking back into range.  Replaces randperm + gather: the kernel reads row perm(i).
struct Shuffle {
    uint64_t n;
    uint32_t abits, bbits;  // left / right half widths, abits + bbits = ceil(log2 n) (>= 2)
    uint32_t key[4];
    int on;
};

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}

__device__ __forceinline__ uint64_t shuffle_index(const Shuffle& s, uint64_t i)
{
    if (!s.on) return i;
    const uint32_t ma = (1u << s.abits) - 1u, mb = (1u << s.bbits) - 1u;  // widths <= 31
    do {
        uint32_t L = (uint32_t)(i >> s.bbits) & ma, R = (uint32_t)i & mb;
        // (L:a, R:b) -> (R:b, L ^ F(R):a) -> ... ; after 4 rounds the widths are (a, b) again
        uint32_t t;
        t = L ^ (mix32(R ^ s.key[0]) & ma); L = R; R = t;  // now L:b R:a
        t = L ^ (mix32(R ^ s.key[1]) & mb); L = R; R = t;  // now L:a R:b
        t = L ^ (mix32(R ^ s.key[2]) & ma); L = R; R = t;
        t = L ^ (mix32(R ^ s.key[3]) & mb); L = R; R = t;
        i = ((uint64_t)L << s.bbits) | R;
    } while (i >= s.n);
    return i;
}

struct MlpTrainArgs {
    const float* data;    // [n][8]: 7 normalised features + normalised target (the whole epoch)
    const float* params;  // [train_params(L)]
    float* partial;       // [gridDim.x][g_stride(L)]: gradient sums + batch loss
    int64_t row0, nrows;  // this step's minibatch = epoch positions row0 .. row0 + nrows
    Shuffle shuf;         // epoch position -> stored row
    int ntiles;
    float two_over_b, inv_keep;
    uint32_t keep16, step, k0, k1;
    // sharded training (omc_mlp_train_epoch_sharded): dropout key of local row i of this step = its position in the
    // GLOBAL minibatch (so every rank draws the masks of the unsharded run); null: the row's own position
    const uint32_t* drop_pos = nullptr;
};

// sizes for L hidden layers of 64 units (flat layout: mlp_params_of(64, L))
__host__ __device__ constexpr int train_params(int L) { return kH * 8 + (L - 1) * (kH * kH + kH) + kH + 1; }
__host__ __device__ constexpr int g_stride(int L) { return (train_params(L) + 1 + 63) / 64 * 64; }
__host__ __device__ constexpr int lds_weights(int L) { return kH * kLdW1 + (L - 1) * (kH * kLdW2 + kH) + kH + 4; }
__host__ __device__ constexpr int lds_wave(int L) { return L * kH * 32 + 9 * 32; }  // L patches + X + dout row
__host__ __device__ constexpr int train_lds_floats(int L)
{
    return lds_weights(L) + 4 * lds_wave(L) > 4 * g_stride(L) ? lds_weights(L) + 4 * lds_wave(L) : 4 * g_stride(L);
}
static_assert(g_stride(2) == kMlpPartialStride2 && g_stride(3) == kMlpPartialStride3, "omc_kernels.h");
static_assert(train_lds_floats(3) * 4 <= 160 * 1024, "LDS of one CU");

// L hidden layers.  Connection j (j = 1 .. L-1) is the 64x64 matrix from hidden layer j-1 to j;
// patch j holds H_j until its weight-gradient product has read it, then dZ_j.
template <int L>
__global__ __launch_bounds__(256) void mlp_train_kernel(MlpTrainArgs a)
{
    constexpr int kConn = kH * kLdW2 + kH;  // one staged connection: weights [64][65] + bias [64]
    extern __shared__ float lds[];
    float* sW1 = lds;
    float* sWc = sW1 + kH * kLdW1;            // (L-1) connections, pre-multiplied by 1 / keep
    float* sWo = sWc + (L - 1) * kConn;       // output weights (pre-multiplied) ...
    float* sBo = sWo + kH;                    // ... and bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    float* patch = lds + lds_weights(L) + wave * lds_wave(L);  // [L][64][32]
    float* tX = patch + L * kH * 32;                           // [8][32] inputs + [32] d(loss)/d(out)

    const StageOfs so = stage_offsets(c, h);
    const int nwaves = gridDim.x * 4;
    auto fetch = [&](int tile) {
        const int64_t r = (int64_t)tile * 32 + c;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (tile < a.ntiles && r < a.nrows)
            v = reinterpret_cast<const float4*>(a.data + shuffle_index(a.shuf, (uint64_t)(a.row0 + r)) * 8)[h];
        return v;
    };
    float4 xnext = fetch(blockIdx.x * 4 + wave);  // travels while the weights are staged

    // Inverted dropout's 1 / keep is folded into the weights that consume the dropped activations
    // (connections and output weights are staged pre-multiplied), so activations carry the 0/1
    // mask only: with H' = mask * relu(Z), Ws = W / keep: Z_j = Ws H'_{j-1} + b, dZ_{j-1} =
    // [H'_{j-1} > 0] Ws^T dZ_j, and the gradients taken against H' (connections, output weights)
    // are multiplied by 1 / keep once, on the way out.
    // Every load of the staging is issued before the first LDS store (one global round trip for the whole parameter
    // block instead of one per pair of loop iterations: the launch's fixed cost, 11.9 -> see DESIGN.md section 6).
    {
        constexpr int kTail = kH * 8 + (L - 1) * (kH * kH + kH);  // output weights [kH], then the output bias
        float w1v[2], wcv[L - 1][16], bcv[L - 1];
#pragma unroll
        for (int k = 0; k < 2; ++k) w1v[k] = a.params[tid + 256 * k];  // kH * 8 = 512
#pragma unroll
        for (int j = 0; j < L - 1; ++j) {
            const float* src = a.params + kH * 8 + j * (kH * kH + kH);
#pragma unroll
            for (int k = 0; k < 16; ++k) wcv[j][k] = src[tid + 256 * k];  // kH * kH = 4096
            bcv[j] = src[kH * kH + (tid & (kH - 1))];
        }
        float wov = a.params[kTail + (tid & (kH - 1))], bov = a.params[kTail + kH];
        asm volatile("" : "+v"(wov), "+v"(bov));  // keeps these two loads in the batch (hipcc sinks them below the stores)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            sW1[(i >> 3) * kLdW1 + (i & 7)] = w1v[k];
        }
#pragma unroll
        for (int j = 0; j < L - 1; ++j) {
            float* dst = sWc + j * kConn;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = tid + 256 * k;
                dst[(i >> 6) * kLdW2 + (i & 63)] = wcv[j][k] * a.inv_keep;
            }
            if (tid < kH) dst[kH * kLdW2 + tid] = bcv[j];
        }
        if (tid < kH) sWo[tid] = wov * a.inv_keep;
        if (tid == 0) sBo[0] = bov;
    }
    __syncthreads();

    v16f gW[L - 1][2][2];
    v4f gW1[4];  // 16x16 tiles: units 16*mt4 + 4*(lane/16) + r, input column lane%16 (< 8 exist)
    float gb[L - 1][2], gwo[2] = {0.0f, 0.0f}, gbo = 0.0f, loss = 0.0f;
#pragma unroll
    for (int j = 0; j < L - 1; ++j) {
        gb[j][0] = gb[j][1] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) gW[j][i][0][r] = gW[j][i][1][r] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) gW1[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    // L == 2 (config 5's net): the one 64 x 64 connection also lives in REGISTERS, in the two operand layouts the
    // tile loop consumes -- forward A operand W[unit c / 32 + c][k-step's unit] and its transpose for dH -- 128 of
    // the 148 registers this instantiation leaves free (one wave per SIMD: 512 are there).  That removes 128 of a
    // tile's ~200 LDS operand reads and the waits in front of the MFMAs that consume them.  (L == 3 has no room.)
    constexpr bool kRegW = (L == 2);
    float wfw[kRegW ? 2 : 1][kRegW ? 2 : 1][kRegW ? 16 : 1], wbw[kRegW ? 2 : 1][kRegW ? 2 : 1][kRegW ? 16 : 1];
    if constexpr (kRegW) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = unit_of(kt, s, h);
                wfw[0][kt][s] = sWc[(c)*kLdW2 + k];
                wfw[1][kt][s] = sWc[(32 + c) * kLdW2 + k];
                wbw[0][kt][s] = sWc[k * kLdW2 + c];
                wbw[1][kt][s] = sWc[k * kLdW2 + 32 + c];
            }
    }

    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += nwaves) {
        const int64_t row = (int64_t)tile * 32 + c;
        const uint32_t drow = (a.drop_pos && a.keep16 < 65536u && row < a.nrows) ? a.drop_pos[row] : (uint32_t)row;
        const bool live = row < a.nrows;
        float4 x = xnext;
        float y = x.w;                 // upper half-wave: column 7 is the target ...
        if (h == 1) x.w = 1.0f;        // ... and its slot carries the bias input
        y = __shfl(y, c + 32, 64);
        wave_sync_lds();               // the previous tile's staging reads are done
        tX[st_idx(4 * h + 0, c)] = x.x;
        tX[st_idx(4 * h + 1, c)] = x.y;
        tX[st_idx(4 * h + 2, c)] = x.z;
        tX[st_idx(4 * h + 3, c)] = x.w;

        // ---- layer 0: Z^T [64 x 32] = W1a [64 x 8] * Xa^T [8 x 32]; k-step s <-> inputs s, s+4
        v16f act[L][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) act[0][mt][r] = 0.0f;
            const float* wr = sW1 + (32 * mt + c) * kLdW1 + 4 * h;
            act[0][mt] = mfma(wr[0], x.x, act[0][mt]);
            act[0][mt] = mfma(wr[1], x.y, act[0][mt]);
            act[0][mt] = mfma(wr[2], x.z, act[0][mt]);
            act[0][mt] = mfma(wr[3], x.w, act[0][mt]);
        }
        // next tile's rows: issued once this tile's are consumed (a wait on the older load would
        // otherwise drain this one too), in flight for the rest of the tile
        xnext = fetch(tile + nwaves);
        relu_dropout<false>(act[0], drow, a.step, 0x100u + (uint32_t)h, a.keep16, a.inv_keep, a.k0, a.k1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[unit_base(mt, r) + so.wr[wr_sel(r)]] = act[0][mt][r];

        // ---- layers 1 .. L-1: Z_j^T = Ws_j * H_{j-1}^T + b_j; k-step (kt, s) <-> unit_of(kt, s, h)
#pragma unroll
        for (int j = 1; j < L; ++j) {
            const float* W = sWc + (j - 1) * kConn;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[j][mt][r] = W[kH * kLdW2 + unit_of(mt, r, h)];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int k = unit_of(kt, s, h);
                    if constexpr (kRegW) {
                        act[j][0] = mfma(wfw[0][kt][s], act[j - 1][kt][s], act[j][0]);
                        act[j][1] = mfma(wfw[1][kt][s], act[j - 1][kt][s], act[j][1]);
                    } else {
                        act[j][0] = mfma(W[(c)*kLdW2 + k], act[j - 1][kt][s], act[j][0]);
                        act[j][1] = mfma(W[(32 + c) * kLdW2 + k], act[j - 1][kt][s], act[j][1]);
                    }
                }
            }
            relu_dropout<false>(act[j], drow, a.step, 0x100u * (uint32_t)(j + 1) + (uint32_t)h, a.keep16,
                                a.inv_keep, a.k0, a.k1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[j * kH * 32 + unit_base(mt, r) + so.wr[wr_sel(r)]] = act[j][mt][r];
        }

        // ---- output, loss, d(loss)/d(out)
        float o = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o = __builtin_fmaf(sWo[unit_of(mt, r, h)], act[L - 1][mt][r], o);
        o += __shfl_xor(o, 32, 64);
        o += sBo[0];
        const float diff = live ? o - y : 0.0f;
        const float dout = diff * a.two_over_b;
        if (h == 0) {
            loss = __builtin_fmaf(diff, diff, loss);
            gbo += dout;
            tX[8 * 32 + c] = dout;
        }
        // ---- back through the last activation: dz = dZ_{L-1}, B operand of the first dH product
        v16f dz[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dz[mt][r] = act[L - 1][mt][r] > 0.0f ? sWo[unit_of(mt, r, h)] * dout : 0.0f;
        wave_sync_lds();

#pragma unroll
        for (int j = L - 1; j >= 1; --j) {
            const float* W = sWc + (j - 1) * kConn;
            const float* pA = patch + j * kH * 32;        // H_j (j = L-1: dZ_j is rebuilt from it) or dZ_j
            const float* pB = patch + (j - 1) * kH * 32;  // H_{j-1}
            // ---- gW_j [i][k] += sum_rows dZ_j[i][row] * H_{j-1}[k][row]; k-step <-> rows 16h + 4q + jj.
            // This lane owns units c and 32 + c here.  For the last hidden layer dZ of those units is
            // rebuilt from H (dZ = [H > 0] wo dout) and the output-weight gradient sum_rows dout * H
            // comes along; bias gradients are the row sums of dZ.
            const float wo0 = sWo[c], wo1 = sWo[32 + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 e0 = *reinterpret_cast<const float4*>(pA + so.rd[q]);
                const float4 e1 = *reinterpret_cast<const float4*>(pA + 32 * 32 + so.rd[q]);
                const float4 dq = *reinterpret_cast<const float4*>(tX + 8 * 32 + 16 * h + 4 * q);
                const float4 b0 = *reinterpret_cast<const float4*>(pB + so.rd[q]);
                const float4 b1 = *reinterpret_cast<const float4*>(pB + 32 * 32 + so.rd[q]);
                const float ev0[4] = {e0.x, e0.y, e0.z, e0.w}, ev1[4] = {e1.x, e1.y, e1.z, e1.w};
                const float dv[4] = {dq.x, dq.y, dq.z, dq.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float av0 = ev0[jj], av1 = ev1[jj];
                    if (j == L - 1) {
                        gwo[0] = __builtin_fmaf(dv[jj], ev0[jj], gwo[0]);
                        gwo[1] = __builtin_fmaf(dv[jj], ev1[jj], gwo[1]);
                        av0 = ev0[jj] > 0.0f ? wo0 * dv[jj] : 0.0f;
                        av1 = ev1[jj] > 0.0f ? wo1 * dv[jj] : 0.0f;
                    }
                    gb[j - 1][0] += av0;
                    gb[j - 1][1] += av1;
                    gW[j - 1][0][0] = mfma(av0, bv0[jj], gW[j - 1][0][0]);
                    gW[j - 1][0][1] = mfma(av0, bv1[jj], gW[j - 1][0][1]);
                    gW[j - 1][1][0] = mfma(av1, bv0[jj], gW[j - 1][1][0]);
                    gW[j - 1][1][1] = mfma(av1, bv1[jj], gW[j - 1][1][1]);
                }
            }
            // ---- dH_{j-1}^T = Ws_j^T * dZ_j^T, then through layer j-1's activation
            v16f d[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) d[mt][r] = 0.0f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if constexpr (kRegW) {
                        d[0] = mfma(wbw[0][it][s], dz[it][s], d[0]);
                        d[1] = mfma(wbw[1][it][s], dz[it][s], d[1]);
                    } else {
                        const float* wr = W + unit_of(it, s, h) * kLdW2 + c;
                        d[0] = mfma(wr[0], dz[it][s], d[0]);
                        d[1] = mfma(wr[32], dz[it][s], d[1]);
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[mt][r] = act[j - 1][mt][r] > 0.0f ? d[mt][r] : 0.0f;
            wave_sync_lds();  // the product above has read patch j-1 (H_{j-1})
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[(j - 1) * kH * 32 + unit_base(mt, r) + so.wr[wr_sel(r)]] = dz[mt][r];
            wave_sync_lds();
        }

        // ---- gW1a [i][n] += sum_rows dZ_0[i][row] * Xa[n][row]; only input columns n < 8 exist, so
        // this product runs on 16x16x4 tiles (unit = lane%16 + 16*mt4, column = lane%16); k-slot
        // kq = lane/16 of step s stands for row 8*kq + s: a lane reads its 8 rows as two quads
        {
            const int l16 = lane & 15, kq = lane >> 4, sw = ((l16 >> 1) & 7) << 2;
            const int o0 = l16 * 32 + ((8 * kq) ^ sw), o1 = l16 * 32 + ((8 * kq + 4) ^ sw);
            float4 bq0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), bq1 = bq0;
            if (l16 < 8) {
                bq0 = *reinterpret_cast<const float4*>(tX + o0);
                bq1 = *reinterpret_cast<const float4*>(tX + o1);
            }
            const float bv[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
            float av[4][8];
#pragma unroll
            for (int mt4 = 0; mt4 < 4; ++mt4) {
                const float4 aq0 = *reinterpret_cast<const float4*>(patch + 16 * 32 * mt4 + o0);
                const float4 aq1 = *reinterpret_cast<const float4*>(patch + 16 * 32 * mt4 + o1);
                av[mt4][0] = aq0.x; av[mt4][1] = aq0.y; av[mt4][2] = aq0.z; av[mt4][3] = aq0.w;
                av[mt4][4] = aq1.x; av[mt4][5] = aq1.y; av[mt4][6] = aq1.z; av[mt4][7] = aq1.w;
            }
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8)  // the four tiles take turns: no MFMA waits for the one before it
#pragma unroll
                for (int mt4 = 0; mt4 < 4; ++mt4)
                    gW1[mt4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt4][s8], bv[s8], gW1[mt4], 0, 0, 0);
        }
    }

    // ---- per-lane partials -> per-wave sums (a lane holds the rows of its half-wave only)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        gwo[mt] += __shfl_xor(gwo[mt], 32, 64);
#pragma unroll
        for (int j = 0; j < L - 1; ++j) gb[j][mt] += __shfl_xor(gb[j][mt], 32, 64);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        gbo += __shfl_xor(gbo, m, 64);
        loss += __shfl_xor(loss, m, 64);
    }

    // ---- the whole LDS allocation (weights are no longer needed) is re-cut into one gradient
    // vector per wave: plain stores, nothing to wait for; then the workgroup adds the copies in
    // wave order on the way out
    constexpr int GS = g_stride(L), NP = train_params(L);
    __syncthreads();
    float* G = lds + wave * GS;
#pragma unroll
    for (int j = 0; j < L - 1; ++j) {
        float* Gj = G + kH * 8 + j * (kH * kH + kH);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = unit_of(mi, r, h);
                Gj[i * kH + c] = gW[j][mi][0][r] * a.inv_keep;
                Gj[i * kH + 32 + c] = gW[j][mi][1][r] * a.inv_keep;
            }
            if (h == 0) Gj[kH * kH + 32 * mi + c] = gb[j][mi];
        }
    }
    if (h == 0) {
        G[NP - 1 - kH + c] = gwo[0] * a.inv_keep;
        G[NP - 1 - kH + 32 + c] = gwo[1] * a.inv_keep;
    }
    if ((lane & 15) < 8) {
#pragma unroll
        for (int mt4 = 0; mt4 < 4; ++mt4)
#pragma unroll
            for (int r = 0; r < 4; ++r) G[(16 * mt4 + 4 * (lane >> 4) + r) * 8 + (lane & 15)] = gW1[mt4][r];
    }
    if (lane == 0) {
        G[NP - 1] = gbo;
        G[NP] = loss;
    }
    __syncthreads();
    float* out = a.partial + (size_t)blockIdx.x * GS;
    for (int i = tid; i <= NP; i += 256) out[i] = ((lds[i] + lds[GS + i]) + lds[2 * GS + i]) + lds[3 * GS + i];
}

struct MlpAdamArgs {
    float* params;
    float* m;
    float* v;
    const float* partial;
    double* loss_acc;  // running sum of batch-mean losses of the epoch
    int nparts, nparams, stride;  // the loss slot is index nparams
    float inv_b, lr_t, inv_sqrt_bc2, beta1, beta2, eps, wd;
    float* wt;  // optional transposed copies of the connections (tile-per-wave trainer), else null
    int H, L;
};

// 16 parameters per workgroup, 16 threads per parameter: thread (slice, j) sums partials
// slice, slice + 16, ... of parameter j (independent loads, all in flight together), the 16 slice
// sums are added in slice order through LDS, thread (0, j) applies Adam.
// (Measured in round 4: 64 parameters per workgroup -- 256-byte reads of a partial -- takes the same 5.0 us at 256
// partials; the kernel is bound by the latency of its 16 loads per thread, not by their coalescing.)
__device__ __forceinline__ void mlp_adam_body(const MlpAdamArgs& a)
{
    __shared__ float red[16][17];
    const int j = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int p = blockIdx.x * 16 + j;
    // the parameter and its moments travel together with the partials (one memory latency per launch, not two)
    float w0 = 0.0f, m0 = 0.0f, v0 = 0.0f;
    if (slice == 0 && p < a.nparams) {
        w0 = a.params[p];
        m0 = a.m[p];
        v0 = a.v[p];
    }
    float g = 0.0f;
    if (p <= a.nparams) {
#pragma unroll 16
        for (int w = slice; w < a.nparts; w += 16) g += a.partial[(size_t)w * a.stride + p];
    }
    red[slice][j] = g;
    __syncthreads();
    if (slice != 0 || p > a.nparams) return;
    g = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) g += red[s2][j];
    if (p == a.nparams) {  // the loss slot
        *a.loss_acc += (double)g * (double)a.inv_b;
        return;
    }
    g = __builtin_fmaf(a.wd, w0, g);
    const float m = __builtin_fmaf(a.beta1, m0, (1.0f - a.beta1) * g);
    const float v = __builtin_fmaf(a.beta2, v0, (1.0f - a.beta2) * g * g);
    a.m[p] = m;
    a.v[p] = v;
    const float denom = __builtin_amdgcn_sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    const float w1 = w0 - a.lr_t * (m / denom);
    a.params[p] = w1;
    if (a.wt) {
        const int conn = a.H * a.H + a.H, q = p - a.H * 8;
        if (q >= 0 && q < (a.L - 1) * conn) {
            const int jc = q / conn, rem = q - jc * conn;
            if (rem < a.H * a.H) {
                const int i = rem / a.H, k = rem - i * a.H;
                a.wt[(size_t)jc * a.H * a.H + (size_t)k * a.H + i] = w1;
            }
        }
    }
}

// The same update with ONE THREAD per parameter (256 parameters per workgroup): for the batched launches, where a
// workgroup of mlp_adam_body per 16 parameters and problem means tens of thousands of nearly empty workgroups.  The
// float additions are those of mlp_adam_body in the same order -- slice sums g_s = partial[s] + partial[s + 16] + ...
// (from 0), then g_0 + g_1 + ... + g_15 (from 0) -- so the parameters come out bit for bit the same.
__device__ __forceinline__ void mlp_adam_body_flat(const MlpAdamArgs& a, const int p)
{
    if (p > a.nparams) return;
    float g = 0.0f;
#pragma unroll 1
    for (int s0 = 0; s0 < 16; ++s0) {
        float gs = 0.0f;
        for (int w = s0; w < a.nparts; w += 16) gs += a.partial[(size_t)w * a.stride + p];
        g += gs;
    }
    if (p == a.nparams) {  // the loss slot
        *a.loss_acc += (double)g * (double)a.inv_b;
        return;
    }
    const float w0 = a.params[p];
    g = __builtin_fmaf(a.wd, w0, g);
    const float m = __builtin_fmaf(a.beta1, a.m[p], (1.0f - a.beta1) * g);
    const float v = __builtin_fmaf(a.beta2, a.v[p], (1.0f - a.beta2) * g * g);
    a.m[p] = m;
    a.v[p] = v;
    const float denom = __builtin_amdgcn_sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    const float w1 = w0 - a.lr_t * (m / denom);
    a.params[p] = w1;
    if (a.wt) {
        const int conn = a.H * a.H + a.H, q = p - a.H * 8;
        if (q >= 0 && q < (a.L - 1) * conn) {
            const int jc = q / conn, rem = q - jc * conn;
            if (rem < a.H * a.H) {
                const int i = rem / a.H, k = rem - i * a.H;
                a.wt[(size_t)jc * a.H * a.H + (size_t)k * a.H + i] = w1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void mlp_adam_kernel(MlpAdamArgs a)
{
    mlp_adam_body(a);
}

// Sharded training.  (1) this rank's gradient sums of the step, partials added in mlp_adam_body's order, as doubles
// (the loss sum rides in slot nparams) -> (2) the host enqueues the all-reduce over the ranks -> (3) Adam from the
// reduced sums: mlp_adam_body with ONE partial, the float of the global sum.
__global__ __launch_bounds__(256) void mlp_grad_reduce_kernel(const float* __restrict__ partial, int nparts, int stride,
                                                              int nparams, double* __restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p > nparams) return;
    float g = 0.0f;
#pragma unroll 1
    for (int s0 = 0; s0 < 16; ++s0) {
        float gs = 0.0f;
        for (int w = s0; w < nparts; w += 16) gs += partial[(size_t)w * stride + p];
        g += gs;
    }
    out[p] = (double)g;
}

__global__ __launch_bounds__(256) void mlp_grad_to_float_kernel(const double* __restrict__ in, int n, float* __restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < n) out[p] = (float)in[p];
}

// ------------------------------------------------------------------ feature statistics
// Means and population variances of the six non-constant regression features
// [x, x^2, x^3, max(x-1,0), s, x*s] (create_regression_features, options_model_3.py:105-121;
// s = sqrt(max(T - t*dt, 1e-6))) and of the target over all R rows, in float64
// (:550-563).  PASS 0 sums values, PASS 1 sums squared deviations from the given means.
struct StatArgs {
    const double* x;
    const int32_t* t;
    const double* y;
    int64_t n;
    double T, dt;
    const double* mean;  // [8] (PASS 1)
    double* part;        // [8][pstride]
    int pstride;
};

template <int PASS>
__global__ __launch_bounds__(kBlock) void nn_stats_kernel(StatArgs a)
{
    __shared__ double red[kNQ * kRedStride];
    double acc[8], mu[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        acc[q] = 0.0;
        mu[q] = PASS ? a.mean[q] : 0.0;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        const double x = a.x[i];
        const double s = sqrt(fmax(a.T - (double)a.t[i] * a.dt, 1e-6));
        const double x2 = x * x;
        const double f[8] = {x, x2, x2 * x, fmax(x - 1.0, 0.0), s, x * s, a.y[i], 0.0};
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const double d = f[q] - mu[q];
            acc[q] += PASS ? d * d : d;
        }
    }
    const double r = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0)
        a.part[(size_t)(threadIdx.x >> 3) * a.pstride + blockIdx.x] = r;
}

// part[q][0..nblk) summed in index order, divided by n -> out[q]
__global__ __launch_bounds__(kBlock) void nn_stats_finish_kernel(const double* part, int nblk, int pstride,
                                                                double n, double* out)
{
    __shared__ double red[kNQ * kRedStride];
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += part[(size_t)q * pstride + i];
    }
    const double r = block_reduce8(acc, red);
    if (threadIdx.x < 64 && (threadIdx.x & 7) == 0) out[threadIdx.x >> 3] = r / n;
}

// ------------------------------------------------------------------ pass 2 with the network
// Sticky backward sweep of the NN flow (options_model_3.py:615-649): at every step the
// continuation value of every still-alive in-the-money path is the network's output on the
// normalised features; exercise where payoff > continuation (strict), first hit going backwards
// sticks.  One wave owns 32 paths for the whole sweep (state in registers), the forward pass
// is the training kernel's (float32 MFMA, transposed layout), dropout stays ACTIVE when asked
// for (the reference never switches the net to eval mode).  Leaves (sx, tex)
// for the common valuation kernel.
struct MlpApplyArgs {
    const float* S;
    int64_t ld, M;
    int N, is_put;
    double K, T, dt;
    const float* params;
    double fm[7], rs[7];  // feature means, reciprocal stds
    double ym, ysd;
    float* sx;
    int32_t* tex;
    float inv_keep;
    uint32_t keep16, k0, k1;
    int ntiles;
    // dropout key of column p: its column in the UNSHARDED matrix (p + base0 for the first half of this matrix's
    // columns, p - half + base1 for the second), so that a shard draws the masks the single GPU draws
    int64_t key_half, key_base0, key_base1;
};

// Flat parameter layout for any (H hidden units, L hidden layers): W1|b1 as [H][8], then for
// every further hidden layer its weights [H][H] and bias [H], then the output weights [H] and
// bias [1].  (64, 2) is the trainer's layout above.
__host__ __device__ constexpr int mlp_params_of(int H, int L) { return H * 8 + (L - 1) * (H * H + H) + H + 1; }
__host__ __device__ constexpr int apply_lds_floats(int H, int L)
{
    return H * kLdW1 + (L - 1) * (H * (H + 1) + H) + H + 4;
}

// ReLU + inverted dropout over NT 32-unit tiles; one Philox block seeds the two streams of a
// tile pair.
__device__ __forceinline__ void relu_dropout_1(v16f& z, uint32_t row, uint32_t step, uint32_t tag, uint32_t keep16,
                                               float inv_keep, uint32_t k0, uint32_t k1);
template <int NT>
__device__ __forceinline__ void relu_dropout_n(v16f (&z)[NT], uint32_t row, uint32_t step, uint32_t tag,
                                               uint32_t keep16, float inv_keep, uint32_t k0, uint32_t k1)
{
    if constexpr (NT == 1) {  // 32 hidden units: one tile, no partner -- the single-stream generator (relu_dropout_1)
        relu_dropout_1(z[0], row, step, tag, keep16, inv_keep, k0, k1);
        return;
    }
#pragma unroll
    for (int p = 0; p + 1 < NT; p += 2) {
        v16f pair[2] = {z[p], z[p + 1]};
        relu_dropout<true>(pair, row, step, tag + 0x1000u * (uint32_t)p, keep16, inv_keep, k0, k1);
        z[p] = pair[0];
        z[p + 1] = pair[1];
    }
}

// ------------------------------------------------------------------ tile-per-wave trainer
// The same training step for the shapes and batch sizes the workgroup kernel above does not fit:
// 128 hidden units (two 128 x 128 matrices do not fit LDS beside the staging patches) and small
// minibatches (the reference's own batch of 256 rows is 8 tiles).  One wave = one 32-row tile =
// one workgroup; the A operands (weights) come straight from global memory / L2 -- tiles are few,
// so that traffic is small -- and because a wave sees exactly one tile there is nothing to
// accumulate across tiles: each 32 x H strip of a weight gradient is complete after its 16 k-steps
// and goes directly to the tile's gradient partial in global memory.
//
// Units are dealt to the 32-unit MFMA tiles round-robin: slot 32*mt + rho <-> unit NT*rho + mt
// (NT = H / 32).  A lane that owns tile-row rho = c then needs, for one k, the NT consecutive
// weights W[k][NT*c .. NT*c + NT-1]: one 16-byte load feeds NT MFMAs, and the NT accumulators of a
// gradient strip store as one 16-byte chunk per register.  LDS staging is addressed by slot (same
// patches and swizzle as above), global parameters and gradients by unit.  Forward products read a
// TRANSPOSED copy of each connection (kept up to date by the Adam kernel) so that those loads are
// row-contiguous too.
struct MlpTileArgs {
    const float* data;
    const float* params;  // canonical layout, mlp_params_of(H, L)
    const float* wt;      // (L-1) x [H][H]: connection j transposed, wt_j[k][i] = W_j[i][k]
    float* partial;       // [ntiles][pstride]
    int64_t row0, nrows;
    Shuffle shuf;
    int ntiles, pstride;
    int tile0, accumulate;  // this launch: tiles tile0 .. tile0 + gridDim.x; add onto the partials?
    float two_over_b, inv_keep;
    uint32_t keep16, step, k0, k1;
    const uint32_t* drop_pos = nullptr;  // see MlpTrainArgs
};

template <int NT>
struct VecN;
template <>
struct VecN<2> { typedef float2 type; };
template <>
struct VecN<4> { typedef float4 type; };
template <int NT>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[NT])
{
    const typename VecN<NT>::type x = *reinterpret_cast<const typename VecN<NT>::type*>(p);
    v[0] = x.x; v[1] = x.y;
    if constexpr (NT == 4) { v[2] = x.z; v[3] = x.w; }
}
template <int NT>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[NT])
{
    typename VecN<NT>::type x;
    x.x = v[0]; x.y = v[1];
    if constexpr (NT == 4) { x.z = v[2]; x.w = v[3]; }
    *reinterpret_cast<typename VecN<NT>::type*>(p) = x;
}

template <int NT, bool SCALE>
__device__ __forceinline__ void relu_dropout_t(v16f (&z)[NT], uint32_t row, uint32_t step, uint32_t tag,
                                               uint32_t keep16, float inv_keep, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int p = 0; p < NT; p += 2) {
        v16f pair[2] = {z[p], z[p + 1]};
        relu_dropout<SCALE>(pair, row, step, tag + 0x1000u * (uint32_t)p, keep16, inv_keep, k0, k1);
        z[p] = pair[0];
        z[p + 1] = pair[1];
    }
}

// concurrent waves (= gradient partials) of the tile-per-wave trainer: what fits the chip at once
__host__ __device__ constexpr int tile_waves_max(int H) { return H == 128 ? 768 : 1024; }
__host__ __device__ constexpr int tile_lds_floats(int H, int L) { return L * H * 32 + 9 * 32 + (L - 1) * H + H + 4; }
__host__ __device__ constexpr int tile_pstride(int H, int L) { return (mlp_params_of(H, L) + 1 + 63) / 64 * 64; }

template <int H, int L>
__global__ __launch_bounds__(64) void mlp_train_tile_kernel(MlpTileArgs a)
{
    constexpr int NT = H / 32, NP = mlp_params_of(H, L), CONN = H * H + H;
    constexpr int kPF = 4;  // k-steps per prefetch group
    extern __shared__ float lds[];
    float* patch = lds;                    // [L][H slots][32 rows]
    float* tX = patch + L * H * 32;        // [8][32] inputs + [32] d(loss)/d(out)
    float* sB = tX + 9 * 32;               // (L-1) x [H] biases, by slot
    float* sWo = sB + (L - 1) * H;         // output weights, by slot
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    auto rho = [&](int r) { return (r >> 2) * 8 + 4 * h + (r & 3); };  // tile row of accumulator register r
    const float* Wo = a.params + H * 8 + (L - 1) * CONN;
    for (int sl = lane; sl < H; sl += 64) {
        const int unit = NT * (sl & 31) + (sl >> 5);
#pragma unroll
        for (int j = 0; j < L - 1; ++j) sB[j * H + sl] = a.params[H * 8 + j * CONN + H * H + unit];
        sWo[sl] = Wo[unit];
    }
    const StageOfs so = stage_offsets(c, h);
    float* out = a.partial + (size_t)blockIdx.x * a.pstride;
    // A minibatch of more tiles than fit the chip at once is covered by several launches of this
    // kernel: launch n handles tiles tile0 + blockIdx.x and, from the second launch on
    // (`accumulate`), adds its gradients onto the wave's partial in global memory -- accumulators
    // start from the stored strip instead of zero.
    const int tile = a.tile0 + blockIdx.x;
    if (tile >= a.ntiles) return;
    const bool first = !a.accumulate;
    const int64_t row = (int64_t)tile * 32 + c;
    const uint32_t drow = (a.drop_pos && a.keep16 < 65536u && row < a.nrows) ? a.drop_pos[row] : (uint32_t)row;
    const bool live = row < a.nrows;
    float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) x = reinterpret_cast<const float4*>(a.data + shuffle_index(a.shuf, (uint64_t)(a.row0 + row)) * 8)[h];
    float y = x.w;
    if (h == 1) x.w = 1.0f;
    y = __shfl(y, c + 32, 64);
    tX[st_idx(4 * h + 0, c)] = x.x;
    tX[st_idx(4 * h + 1, c)] = x.y;
    tX[st_idx(4 * h + 2, c)] = x.z;
    tX[st_idx(4 * h + 3, c)] = x.w;

    // ---- layer 0
    v16f act[L][NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) act[0][mt][r] = 0.0f;
        const float4 w = *reinterpret_cast<const float4*>(a.params + (NT * c + mt) * 8 + 4 * h);
        act[0][mt] = mfma(w.x, x.x, act[0][mt]);
        act[0][mt] = mfma(w.y, x.y, act[0][mt]);
        act[0][mt] = mfma(w.z, x.z, act[0][mt]);
        act[0][mt] = mfma(w.w, x.w, act[0][mt]);
    }
    relu_dropout_t<NT, true>(act[0], drow, a.step, 0x100u + (uint32_t)h, a.keep16, a.inv_keep, a.k0, a.k1);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[unit_base(mt, r) + so.wr[wr_sel(r)]] = act[0][mt][r];
    wave_sync_lds();  // biases, inputs

    // ---- layers 1 .. L-1: A = Wt_j[k][NT*c ..], one vector load per k-step feeds the NT tiles
#pragma unroll
    for (int j = 1; j < L; ++j) {
        const float* Wt = a.wt + (size_t)(j - 1) * H * H;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) act[j][mt][r] = sB[(j - 1) * H + 32 * mt + rho(r)];
        // weights of the next group of kPF k-steps are requested before the current group's MFMAs
        // are issued: an L2 round trip is several MFMA groups long
        {
            float w[2][kPF][NT];
            auto fetch_group = [&](int g, float (&dst)[kPF][NT]) {
#pragma unroll
                for (int i = 0; i < kPF; ++i) {
                    const int ks = g * kPF + i, kt = ks >> 4, s = ks & 15;
                    load_vec<NT>(Wt + (size_t)(NT * rho(s) + kt) * H + NT * c, dst[i]);
                }
            };
            fetch_group(0, w[0]);
#pragma unroll
            for (int g = 0; g < NT * 16 / kPF; ++g) {
                if (g + 1 < NT * 16 / kPF) fetch_group(g + 1, w[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < kPF; ++i) {
                    const int ks = g * kPF + i, kt = ks >> 4, s = ks & 15;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
                        act[j][mt] = mfma(w[g & 1][i][mt], act[j - 1][kt][s], act[j][mt]);
                }
            }
        }
        relu_dropout_t<NT, true>(act[j], drow, a.step, 0x100u * (uint32_t)(j + 1) + (uint32_t)h, a.keep16,
                                 a.inv_keep, a.k0, a.k1);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[j * H * 32 + unit_base(mt, r) + so.wr[wr_sel(r)]] = act[j][mt][r];
    }

    // ---- output, loss, d(loss)/d(out)
    float o = 0.0f;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o = __builtin_fmaf(sWo[32 * mt + rho(r)], act[L - 1][mt][r], o);
    o += __shfl_xor(o, 32, 64);
    o += Wo[H];
    const float diff = live ? o - y : 0.0f;
    const float dout = diff * a.two_over_b;
    if (h == 0) tX[8 * 32 + c] = dout;
    v16f dz[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dz[mt][r] = act[L - 1][mt][r] > 0.0f ? sWo[32 * mt + rho(r)] * dout * a.inv_keep : 0.0f;
    wave_sync_lds();

#pragma unroll
    for (int j = L - 1; j >= 1; --j) {
        const float* W = a.params + H * 8 + (size_t)(j - 1) * CONN;
        float* gWj = out + H * 8 + (size_t)(j - 1) * CONN;
        const float* pA = patch + j * H * 32;
        const float* pB = patch + (j - 1) * H * 32;
        // ---- gW_j strip by strip: rows i = slots 32*mi + c, all H columns; complete after 16 k-steps
#pragma unroll
        for (int mi = 0; mi < NT; ++mi) {
            v16f acc[NT];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v[NT];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) v[ni] = 0.0f;
                if (!first) load_vec<NT>(gWj + (size_t)(NT * rho(r) + mi) * H + NT * c, v);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[ni][r] = v[ni];
            }
            const float woi = sWo[32 * mi + c] * a.inv_keep;
            float gbs = 0.0f, gws = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 e = *reinterpret_cast<const float4*>(pA + 32 * 32 * mi + so.rd[q]);
                const float4 dq = *reinterpret_cast<const float4*>(tX + 8 * 32 + 16 * h + 4 * q);
                const float ev[4] = {e.x, e.y, e.z, e.w}, dv[4] = {dq.x, dq.y, dq.z, dq.w};
                float bv[NT][4];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const float4 b = *reinterpret_cast<const float4*>(pB + 32 * 32 * ni + so.rd[q]);
                    bv[ni][0] = b.x; bv[ni][1] = b.y; bv[ni][2] = b.z; bv[ni][3] = b.w;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float av = ev[jj];
                    if (j == L - 1) {
                        gws = __builtin_fmaf(dv[jj], ev[jj], gws);
                        av = ev[jj] > 0.0f ? woi * dv[jj] : 0.0f;
                    }
                    gbs += av;
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) acc[ni] = mfma(av, bv[ni][jj], acc[ni]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v[NT];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) v[ni] = acc[ni][r];
                store_vec<NT>(gWj + (size_t)(NT * rho(r) + mi) * H + NT * c, v);
            }
            gbs += __shfl_xor(gbs, 32, 64);
            gws += __shfl_xor(gws, 32, 64);
            if (h == 0) {
                float* pb = gWj + H * H + NT * c + mi;
                *pb = first ? gbs : *pb + gbs;
                if (j == L - 1) {
                    float* pw = out + H * 8 + (L - 1) * CONN + NT * c + mi;
                    *pw = first ? gws : *pw + gws;
                }
            }
        }
        // ---- dH_{j-1} = W_j^T dZ_j: A = W_j[i][NT*c ..]
        v16f d[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[mt][r] = 0.0f;
        {
            float w[2][kPF][NT];
            auto fetch_group = [&](int g, float (&dst)[kPF][NT]) {
#pragma unroll
                for (int i = 0; i < kPF; ++i) {
                    const int ks = g * kPF + i, it = ks >> 4, s = ks & 15;
                    load_vec<NT>(W + (size_t)(NT * rho(s) + it) * H + NT * c, dst[i]);
                }
            };
            fetch_group(0, w[0]);
#pragma unroll
            for (int g = 0; g < NT * 16 / kPF; ++g) {
                if (g + 1 < NT * 16 / kPF) fetch_group(g + 1, w[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < kPF; ++i) {
                    const int ks = g * kPF + i, it = ks >> 4, s = ks & 15;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt) d[mt] = mfma(w[g & 1][i][mt], dz[it][s], d[mt]);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[mt][r] = act[j - 1][mt][r] > 0.0f ? d[mt][r] * a.inv_keep : 0.0f;
        wave_sync_lds();
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[(j - 1) * H * 32 + unit_base(mt, r) + so.wr[wr_sel(r)]] = dz[mt][r];
        wave_sync_lds();
    }

    // ---- gW1a: 16x16x4 tiles over slots 16*mt4 + lane%16, k-slot lane/16 of step s <-> row 8*kq + s
    {
        const int l16 = lane & 15, kq = lane >> 4, sw = ((l16 >> 1) & 7) << 2;
        const int o0 = l16 * 32 + ((8 * kq) ^ sw), o1 = l16 * 32 + ((8 * kq + 4) ^ sw);
        float4 bq0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), bq1 = bq0;
        if (l16 < 8) {
            bq0 = *reinterpret_cast<const float4*>(tX + o0);
            bq1 = *reinterpret_cast<const float4*>(tX + o1);
        }
        const float bv[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
#pragma unroll
        for (int mt4 = 0; mt4 < H / 16; ++mt4) {
            const float4 aq0 = *reinterpret_cast<const float4*>(patch + 16 * 32 * mt4 + o0);
            const float4 aq1 = *reinterpret_cast<const float4*>(patch + 16 * 32 * mt4 + o1);
            const float av[8] = {aq0.x, aq0.y, aq0.z, aq0.w, aq1.x, aq1.y, aq1.z, aq1.w};
            v4f g = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) g = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s8], bv[s8], g, 0, 0, 0);
            if (l16 < 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = 16 * mt4 + 4 * (lane >> 4) + r;
                    float* pg = out + (NT * (slot & 31) + (slot >> 5)) * 8 + l16;
                    *pg = first ? g[r] : *pg + g[r];
                }
            }
        }
    }
    float gbo = h == 0 ? dout : 0.0f, loss = h == 0 ? diff * diff : 0.0f;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        gbo += __shfl_xor(gbo, m, 64);
        loss += __shfl_xor(loss, m, 64);
    }
    if (lane == 0) {
        out[NP - 1] = first ? gbo : out[NP - 1] + gbo;
        out[NP] = first ? loss : out[NP] + loss;
    }
}

// ------------------------------------------------------------------ tile-per-workgroup trainer
// Small minibatches -- the reference's own batch of 256 rows is 8 tiles -- leave the tile-per-wave kernel
// with 8 waves on the whole chip, each running ~390 float32 MFMAs of 64 cycles in sequence (65 us per
// step at 3 x 128).  Here one 32-row tile is a WORKGROUP of W = H / 32 waves, one per SIMD: wave w owns
// units 32w .. 32w + 31 of every hidden layer, so each wave issues a quarter of the MFMAs and the four
// SIMDs of the CU work on the tile together.  A layer's activations (and, going back, dZ) are exchanged
// through LDS in the swizzled [unit][32 rows] layout of the other trainers: the forward / dH products
// read them as B operands one value per lane ([unit][row c]: a permutation of a row, conflict-free),
// the weight-gradient products as 16-byte row quads.  Weights come from global memory / L2 (canonical
// layout for W^T dZ, the transposed copy for the forward products: both coalesced over the 32 units a
// wave owns), one group of 16 k-steps ahead of the MFMAs that use them.  Every gradient entry is written
// exactly once per tile (no accumulation across launches): ntiles <= kMlpMaxGroups workgroups, one
// partial each, summed by the Adam kernel in index order -- bitwise reproducible like the others.
struct MlpQuadArgs {
    const float* data;
    const float* params;  // canonical layout, mlp_params_of(H, L)
    const float* wt;      // (L-1) x [H][H]: connection j transposed, wt_j[k][i] = W_j[i][k]
    float* partial;       // [ntiles][pstride]
    int64_t row0, nrows;
    Shuffle shuf;
    int ntiles, pstride;
    float two_over_b, inv_keep;
    uint32_t keep16, step, k0, k1;
    const uint32_t* drop_pos = nullptr;  // see MlpTrainArgs
};

// ReLU + inverted dropout on ONE 32-unit tile's pre-activations (16 per lane), in place; same bit budget as
// relu_dropout (16 bits per unit from a multiply-with-carry stream seeded by one Philox block)
__device__ __forceinline__ void relu_dropout_1(v16f& z, uint32_t row, uint32_t step, uint32_t tag, uint32_t keep16,
                                               float inv_keep, uint32_t k0, uint32_t k1)
{
    if (keep16 >= 65536u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = fmaxf(z[r], 0.0f);
        return;
    }
    const U4 o = philox4x32_10(row, step, tag, 0x4d4c5134u, k0, k1);
    uint64_t st = ((uint64_t)(o.x ^ o.z) << 32) | ((o.y ^ o.w) | 1u);
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        st = (uint64_t)4294957665u * (uint32_t)st + (st >> 32);
        const uint32_t w = (uint32_t)st;
        const float v0 = z[e], v1 = z[e + 1];
        z[e] = (v0 > 0.0f && (w & 0xffffu) < keep16) ? v0 * inv_keep : 0.0f;
        z[e + 1] = (v1 > 0.0f && (w >> 16) < keep16) ? v1 * inv_keep : 0.0f;
    }
}

template <int H, int L>
__device__ __forceinline__ void mlp_train_quad_body(const MlpQuadArgs& a, const int tile)
{
    constexpr int W = H / 32, NP = mlp_params_of(H, L), CONN = H * H + H;
    __shared__ float sAct[L][H * 32];  // H_j, swizzled [unit][32 rows]
    __shared__ float sDz[H * 32];      // dZ_j of the layer being back-propagated
    __shared__ float sX[8 * 32];       // inputs [in][row] (row 7 = the bias column of ones)
    __shared__ float sO[W * 32];       // per-wave partial outputs
    __shared__ float sD[32];           // d(loss)/d(out) per row
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    auto rho = [&](int r) { return (r >> 2) * 8 + 4 * h + (r & 3); };  // tile row of accumulator register r
    if (tile >= a.ntiles) return;
    float* out = a.partial + (size_t)tile * a.pstride;
    const float* Wo = a.params + H * 8 + (L - 1) * CONN;
    const int64_t row = (int64_t)tile * 32 + c;
    const uint32_t drow = (a.drop_pos && a.keep16 < 65536u && row < a.nrows) ? a.drop_pos[row] : (uint32_t)row;
    const bool live = row < a.nrows;
    float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) x = reinterpret_cast<const float4*>(a.data + shuffle_index(a.shuf, (uint64_t)(a.row0 + row)) * 8)[h];
    float y = x.w;
    if (h == 1) x.w = 1.0f;
    y = __shfl(y, c + 32, 64);
    if (w == 0) {
        sX[(4 * h + 0) * 32 + c] = x.x;
        sX[(4 * h + 1) * 32 + c] = x.y;
        sX[(4 * h + 2) * 32 + c] = x.z;
        sX[(4 * h + 3) * 32 + c] = x.w;
    }

    // ---- layer 0: own 32 units x 8 inputs
    v16f hreg[L];
    {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float4 wv = *reinterpret_cast<const float4*>(a.params + (32 * w + c) * 8 + 4 * h);
        acc = mfma(wv.x, x.x, acc);
        acc = mfma(wv.y, x.y, acc);
        acc = mfma(wv.z, x.z, acc);
        acc = mfma(wv.w, x.w, acc);
        relu_dropout_1(acc, drow, a.step, 0x100u + 0x10u * (uint32_t)w + (uint32_t)h, a.keep16, a.inv_keep, a.k0,
                       a.k1);
        hreg[0] = acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sAct[0][st_idx(32 * w + rho(r), c)] = acc[r];
    }
    __syncthreads();

    // 16 k-steps of one 32-unit block kt: A from global memory (row of `Wsrc` = k unit, 32 own units contiguous),
    // B from the swizzled LDS image `Bsrc`; the next block's weights are requested before this block's MFMAs
    auto product = [&](const float* Wsrc, const float* Bsrc, v16f acc) {
        float wa[2][16];
        auto fetch = [&](int kt, float (&dst)[16]) {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) dst[s2] = Wsrc[(size_t)(32 * kt + rho(s2)) * H + 32 * w + c];
        };
        fetch(0, wa[0]);
#pragma unroll
        for (int kt = 0; kt < W; ++kt) {
            if (kt + 1 < W) fetch(kt + 1, wa[(kt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            float b[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) b[s2] = Bsrc[st_idx(32 * kt + rho(s2), c)];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) acc = mfma(wa[kt & 1][s2], b[s2], acc);
        }
        return acc;
    };

    // ---- layers 1 .. L-1
#pragma unroll
    for (int j = 1; j < L; ++j) {
        const float* bj = a.params + H * 8 + (size_t)(j - 1) * CONN + H * H;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bj[32 * w + rho(r)];
        acc = product(a.wt + (size_t)(j - 1) * H * H, sAct[j - 1], acc);
        relu_dropout_1(acc, drow, a.step, 0x100u * (uint32_t)(j + 1) + 0x10u * (uint32_t)w + (uint32_t)h, a.keep16,
                       a.inv_keep, a.k0, a.k1);
        hreg[j] = acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sAct[j][st_idx(32 * w + rho(r), c)] = acc[r];
        __syncthreads();
    }

    // ---- output, loss, d(loss)/d(out): every wave ends with the same numbers
    float wo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wo[r] = Wo[32 * w + rho(r)];
    {
        float o = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) o = __builtin_fmaf(wo[r], hreg[L - 1][r], o);
        o += __shfl_xor(o, 32, 64);
        if (h == 0) sO[w * 32 + c] = o;
    }
    __syncthreads();
    float o = Wo[H];
#pragma unroll
    for (int ww = 0; ww < W; ++ww) o += sO[ww * 32 + c];
    const float diff = live ? o - y : 0.0f;
    const float dout = diff * a.two_over_b;
    if (w == 0 && h == 0) sD[c] = dout;
    v16f dz;
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = hreg[L - 1][r] > 0.0f ? wo[r] * dout * a.inv_keep : 0.0f;
    __syncthreads();

    // swizzled 16-byte read of rows 16h + 4q .. + 3 of `unit`
    auto quad = [&](const float* base, int unit, int q) {
        return *reinterpret_cast<const float4*>(base + unit * 32 + ((16 * h + 4 * q) ^ (((unit >> 1) & 7) << 2)));
    };

    // output-weight gradient of the own units: sum over rows of dout * H_{L-1}
    {
        float gws = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 e = quad(sAct[L - 1], 32 * w + c, q);
            const float4 d = *reinterpret_cast<const float4*>(sD + 16 * h + 4 * q);
            gws = __builtin_fmaf(e.x, d.x, gws);
            gws = __builtin_fmaf(e.y, d.y, gws);
            gws = __builtin_fmaf(e.z, d.z, gws);
            gws = __builtin_fmaf(e.w, d.w, gws);
        }
        gws += __shfl_xor(gws, 32, 64);
        if (h == 0) out[H * 8 + (L - 1) * CONN + 32 * w + c] = gws;
    }

#pragma unroll
    for (int j = L - 1; j >= 1; --j) {
        const float* Wj = a.params + H * 8 + (size_t)(j - 1) * CONN;
        float* gWj = out + H * 8 + (size_t)(j - 1) * CONN;
#pragma unroll
        for (int r = 0; r < 16; ++r) sDz[st_idx(32 * w + rho(r), c)] = dz[r];
        __syncthreads();
        // ---- gW_j rows = own units, all H columns: contraction over the 32 batch rows
        {
            v16f acc[W];
#pragma unroll
            for (int nt = 0; nt < W; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
            float gbs = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 ea = quad(sDz, 32 * w + c, q);
                const float av[4] = {ea.x, ea.y, ea.z, ea.w};
                float bv[W][4];
#pragma unroll
                for (int nt = 0; nt < W; ++nt) {
                    const float4 eb = quad(sAct[j - 1], 32 * nt + c, q);
                    bv[nt][0] = eb.x; bv[nt][1] = eb.y; bv[nt][2] = eb.z; bv[nt][3] = eb.w;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    gbs += av[jj];
#pragma unroll
                    for (int nt = 0; nt < W; ++nt) acc[nt] = mfma(av[jj], bv[nt][jj], acc[nt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < W; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) gWj[(size_t)(32 * w + rho(r)) * H + 32 * nt + c] = acc[nt][r];
            gbs += __shfl_xor(gbs, 32, 64);
            if (h == 0) gWj[H * H + 32 * w + c] = gbs;
        }
        // ---- dH_{j-1} of the own units = W_j^T dZ_j, then through the ReLU / dropout mask of H_{j-1}
        {
            v16f d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.0f;
            d = product(Wj, sDz, d);
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] = hreg[j - 1][r] > 0.0f ? d[r] * a.inv_keep : 0.0f;
        }
        __syncthreads();  // every wave is done with sDz
    }

    // ---- gW1 (own units x 8 inputs, bias in column 7): 16 rows per half-wave on the vector unit
#pragma unroll
    for (int r = 0; r < 16; ++r) sDz[st_idx(32 * w + rho(r), c)] = dz[r];
    __syncthreads();
    {
        float g[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 e = quad(sDz, 32 * w + c, q);
            const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int in = 0; in < 8; ++in) {
                const float4 xv = *reinterpret_cast<const float4*>(sX + in * 32 + 16 * h + 4 * q);
                g[in] = __builtin_fmaf(ev[0], xv.x, g[in]);
                g[in] = __builtin_fmaf(ev[1], xv.y, g[in]);
                g[in] = __builtin_fmaf(ev[2], xv.z, g[in]);
                g[in] = __builtin_fmaf(ev[3], xv.w, g[in]);
            }
        }
#pragma unroll
        for (int in = 0; in < 8; ++in) g[in] += __shfl_xor(g[in], 32, 64);
        if (h == 0) {
            float4* po = reinterpret_cast<float4*>(out + (32 * w + c) * 8);
            po[0] = make_float4(g[0], g[1], g[2], g[3]);
            po[1] = make_float4(g[4], g[5], g[6], g[7]);
        }
    }
    if (w == 0) {
        float gbo = h == 0 ? dout : 0.0f, loss = h == 0 ? diff * diff : 0.0f;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            gbo += __shfl_xor(gbo, m, 64);
            loss += __shfl_xor(loss, m, 64);
        }
        if (lane == 0) {
            out[NP - 1] = gbo;
            out[NP] = loss;
        }
    }
}

template <int H, int L>
__global__ __launch_bounds__(H * 2) void mlp_train_quad_kernel(MlpQuadArgs a)
{
    mlp_train_quad_body<H, L>(a, blockIdx.x);
}

// ------------------------------------------------------------------ 16-row tiles: the reference's own minibatch
// The reference trains on minibatches of min(256, R) rows (options_model_3.py:574): 22,000 optimizer steps of 256 rows
// at its default call.  In 32-row tiles that is 8 workgroups on a 256-CU chip, each running ~390 float32 MFMAs of 64
// cycles behind one another and waiting for every block of weights to come back from L2 (round 5 profile: 23.7 us
// per step at 3 x 128, of which 10 us are MFMA issue).  This kernel halves the serial part and takes the weight
// fetches off the critical path:
//   * a workgroup owns a 16-row tile and runs v_mfma_f32_16x16x4_f32 (the same multiply-adds per cycle as 32x32x2):
//     twice the workgroups, half the MFMA cycles each;
//   * ALL A operands of a 128 x 128 product (64 registers per lane) are requested a whole product ahead -- the forward
//     products' at kernel entry, each backward product's as soon as the forward product that used the same registers
//     is done -- so a product never waits for L2 (one wave per SIMD: 512 registers are there);
//   * activations go through LDS in [k / 4][row][k % 4] order: the B operands of four k-steps are ONE 16-byte read.
// Wave w owns hidden units 32 w .. 32 w + 31 of every layer, as two 16-row MFMA blocks ub = 0, 1 holding the even and
// the odd units (output row m of block ub <-> unit 32 w + 2 m + ub: the two blocks' weights are one 8-byte load).
// After an MFMA lane (row j = lane % 16, g = lane / 16) holds, for its row, the EIGHT CONSECUTIVE units 32 w + 8 g + e,
// e = 2 r + ub (register r of block ub): dropout draws one Philox block (8 x 16 bits) per lane and layer.
// k-step (q, t) of a product contracts k = 16 q + 4 g + t in lane group g -- any order is fine as long as A and B agree.
typedef float v4f16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f16 mfma16(float a, float b, v4f16 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ReLU + inverted dropout on a lane's eight consecutive units (z[ub][r] <-> unit offset e = 2 r + ub): 16 bits per unit
// straight from one Philox block -- word e / 2, half e % 2.  tag = 0x100 * (layer + 1) + (first unit / 8).
__device__ __forceinline__ void relu_dropout_q16(v4f16 (&z)[2], uint32_t row, uint32_t step, uint32_t tag, uint32_t keep16,
                                                 float inv_keep, uint32_t k0, uint32_t k1)
{
    if (keep16 >= 65536u) {
#pragma unroll
        for (int ub = 0; ub < 2; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[ub][r] = fmaxf(z[ub][r], 0.0f);
        return;
    }
    const U4 o = philox4x32_10(row, step, tag, 0x4d4c5138u, k0, k1);
    const uint32_t wd[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v0 = z[0][r], v1 = z[1][r];
        z[0][r] = (v0 > 0.0f && (wd[r] & 0xffffu) < keep16) ? v0 * inv_keep : 0.0f;
        z[1][r] = (v1 > 0.0f && (wd[r] >> 16) < keep16) ? v1 * inv_keep : 0.0f;
    }
}

template <int H, int L>
__device__ __forceinline__ void mlp_train_q16_body(const MlpQuadArgs& a, const int tile)
{
    constexpr int W = H / 32, NP = mlp_params_of(H, L), CONN = H * H + H, NQ = H / 16, KS = H / 4;
    __shared__ __attribute__((aligned(16))) float sAct[L][H * 16];  // H_j, [k / 4][16 rows][k % 4]
    __shared__ __attribute__((aligned(16))) float sDz[H * 16];      // dZ_j of the layer being back-propagated, same order
    __shared__ float sX[8 * 16];                                    // inputs [in][row] (row 7 = the bias column of ones)
    __shared__ float sO[W * 16];                                    // per-wave partial outputs
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, g = lane >> 4;
    if (tile >= a.ntiles) return;
    float* out = a.partial + (size_t)tile * a.pstride;
    const float* Wo = a.params + H * 8 + (L - 1) * CONN;

    // A operands of one H x H product: row k of `src` is the contraction index, this wave's 32 columns 32 w + 2 j, + 1
    auto fetch_w = [&](const float* __restrict__ src, float2 (&dst)[KS]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dst[4 * q + t] = *reinterpret_cast<const float2*>(src + (size_t)(16 * q + 4 * g + t) * H + 32 * w + 2 * j);
            }
    };
    // acc[ub] += A (registers) x B (LDS image of the previous layer / of dZ)
    auto product = [&](const float2 (&wa)[KS], const float* Bsrc, v4f16 (&acc)[2]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(Bsrc + (4 * q + g) * 64 + j * 4);
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = mfma16(wa[4 * q + t].x, bv[t], acc[0]);
                acc[1] = mfma16(wa[4 * q + t].y, bv[t], acc[1]);
            }
        }
    };
    // a lane's eight units of its row -> LDS in [k / 4][row][k % 4] order (two 16-byte stores)
    auto put = [&](float* dst, const v4f16 (&v)[2]) {
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
            *reinterpret_cast<float4*>(dst + (8 * w + 2 * g + rh) * 64 + j * 4) =
                make_float4(v[0][2 * rh], v[1][2 * rh], v[0][2 * rh + 1], v[1][2 * rh + 1]);
    };
    // units 32 w + 2 j, + 1 of an LDS image at row 4 s + g: the operand of the products that contract over the rows
    auto own_pair = [&](const float* src, int s2) {
        return *reinterpret_cast<const float2*>(src + (8 * w + (j >> 1)) * 64 + (4 * s2 + g) * 4 + 2 * (j & 1));
    };

    // ---- every small load first, then the big ones: loads return in order, and each of these would otherwise expose
    // one L2-miss latency on the step's critical path (the parameters were just rewritten by the Adam kernel)
    const int64_t row = (int64_t)tile * 16 + j;
    const bool live = row < a.nrows;
    float xa = 0.0f, xb = 0.0f;  // inputs g and 4 + g of the lane's row
    if (live) {
        const float* xr = a.data + shuffle_index(a.shuf, (uint64_t)(a.row0 + row)) * 8;
        xa = xr[g];
        xb = xr[4 + g];
    }
    const uint32_t drow = (a.drop_pos && a.keep16 < 65536u && live) ? a.drop_pos[row] : (uint32_t)row;
    float w1v[2][2];  // layer-0 weights of units 32 w + 2 j + ub, inputs g and 4 + g
#pragma unroll
    for (int ub = 0; ub < 2; ++ub) {
        const float* wr = a.params + (32 * w + 2 * j + ub) * 8 + g;
        w1v[ub][0] = wr[0];
        w1v[ub][1] = wr[4];
    }
    float4 bias[L > 1 ? L - 1 : 1][2];  // biases of the lane's eight units 32 w + 8 g .. + 7
#pragma unroll
    for (int l = 1; l < L; ++l) {
        const float* bj = a.params + H * 8 + (size_t)(l - 1) * CONN + H * H + 32 * w + 8 * g;
        bias[l - 1][0] = *reinterpret_cast<const float4*>(bj);
        bias[l - 1][1] = *reinterpret_cast<const float4*>(bj + 4);
    }
    const float4 wo0 = *reinterpret_cast<const float4*>(Wo + 32 * w + 8 * g),
                 wo1 = *reinterpret_cast<const float4*>(Wo + 32 * w + 8 * g + 4);
    const float bo = Wo[H];

    float2 wbuf[L - 1][KS];  // connection c: first its transposed copy (forward), then the canonical matrix (dH)
#pragma unroll
    for (int c = 0; c < L - 1; ++c) fetch_w(a.wt + (size_t)c * H * H, wbuf[c]);

    const float y = __shfl(xb, 48 + j, 64);  // column 7 is the target ...
    if (g == 3) xb = 1.0f;                   // ... and its slot carries the bias input
    if (w == 0) {
        sX[g * 16 + j] = xa;
        sX[(4 + g) * 16 + j] = xb;
    }

    // ---- layer 0: own 32 units x 8 inputs, k-steps s = 0, 1 <-> inputs 4 s + g
    v4f16 hreg[L][2];
    {
        v4f16 acc[2];
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            acc[ub] = v4f16{0.0f, 0.0f, 0.0f, 0.0f};
            acc[ub] = mfma16(w1v[ub][0], xa, acc[ub]);
            acc[ub] = mfma16(w1v[ub][1], xb, acc[ub]);
        }
        relu_dropout_q16(acc, drow, a.step, 0x100u + (uint32_t)(4 * w + g), a.keep16, a.inv_keep, a.k0, a.k1);
        hreg[0][0] = acc[0];
        hreg[0][1] = acc[1];
        put(sAct[0], acc);
    }
    __syncthreads();

    // ---- layers 1 .. L-1
#pragma unroll
    for (int l = 1; l < L; ++l) {
        const float4 b0 = bias[l - 1][0], b1 = bias[l - 1][1];
        v4f16 acc[2] = {v4f16{b0.x, b0.z, b1.x, b1.z}, v4f16{b0.y, b0.w, b1.y, b1.w}};  // unit offset e = 2 r + ub
        product(wbuf[l - 1], sAct[l - 1], acc);
        fetch_w(a.params + H * 8 + (size_t)(l - 1) * CONN, wbuf[l - 1]);  // the same connection, canonical: for dH
        relu_dropout_q16(acc, drow, a.step, 0x100u * (uint32_t)(l + 1) + (uint32_t)(4 * w + g), a.keep16, a.inv_keep,
                         a.k0, a.k1);
        hreg[l][0] = acc[0];
        hreg[l][1] = acc[1];
        put(sAct[l], acc);
        __syncthreads();
    }

    // ---- output, loss, d(loss)/d(out): every wave ends with the same numbers
    const float wo[2][4] = {{wo0.x, wo0.z, wo1.x, wo1.z}, {wo0.y, wo0.w, wo1.y, wo1.w}};  // wo[ub][r] <-> unit offset 2 r + ub
    {
        float o = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o = __builtin_fmaf(wo[0][r], hreg[L - 1][0][r], o);
            o = __builtin_fmaf(wo[1][r], hreg[L - 1][1][r], o);
        }
        o += __shfl_xor(o, 16, 64);
        o += __shfl_xor(o, 32, 64);
        if (g == 0) sO[w * 16 + j] = o;
    }
    __syncthreads();
    float o = bo;
#pragma unroll
    for (int ww = 0; ww < W; ++ww) o += sO[ww * 16 + j];
    const float diff = live ? o - y : 0.0f;
    const float dout = diff * a.two_over_b;
    v4f16 dz[2];
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[ub][r] = hreg[L - 1][ub][r] > 0.0f ? wo[ub][r] * dout * a.inv_keep : 0.0f;
    // output-weight gradient of the own units: sum over rows of dout * H_{L-1} as one more contraction over the rows
    // (A = dout of row 4 s + g for every output row, B = H_{L-1} of unit 32 w + 2 n + ub): every output row holds the sum
    {
        v4f16 acc[2] = {v4f16{0.0f, 0.0f, 0.0f, 0.0f}, v4f16{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const float dv = __shfl(dout, 4 * s2 + g, 64);  // lanes 0 .. 15 hold the rows' values
            const float2 hv = own_pair(sAct[L - 1], s2);
            acc[0] = mfma16(dv, hv.x, acc[0]);
            acc[1] = mfma16(dv, hv.y, acc[1]);
        }
        if (g == 0) *reinterpret_cast<float2*>(out + H * 8 + (L - 1) * CONN + 32 * w + 2 * j) = make_float2(acc[0][0], acc[1][0]);
    }
#pragma unroll
    for (int l = L - 1; l >= 1; --l) {
        float* gWl = out + H * 8 + (size_t)(l - 1) * CONN;
        put(sDz, dz);
        __syncthreads();
        // ---- gW_l, transposed product: rows of the MFMA <-> columns k of gW_l (A = H_{l-1} of unit 16 kb + j), columns
        // <-> own units i = 32 w + 2 n + ub (B = dZ_l); contraction over the 16 rows, k-step s <-> rows 4 s + g.  A lane's
        // four registers are gW_l[i][16 kb + 4 g .. + 3]: one 16-byte store.  Block NQ has A = 1: the bias gradient.
        {
            // The column blocks in two halves: half as many accumulators live at a time (with the weights of two
            // connections in flight the kernel sits at the 256 registers that still let two workgroups share a CU --
            // what the side-by-side trainer of a curve's networks needs; every output element sums its four k-steps in
            // the same order either way).  Every LDS operand of a half is requested before its first MFMA (left to
            // itself hipcc puts each ds_read directly in front of the pair of MFMAs that uses it).
            float2 dv[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) dv[s2] = own_pair(sDz, s2);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                constexpr int NH = NQ / 2;
                v4f16 acc[2][NH + 1];
#pragma unroll
                for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                    for (int kb = 0; kb <= NH; ++kb) acc[ub][kb] = v4f16{0.0f, 0.0f, 0.0f, 0.0f};
                float hv[4][NH];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int kb = 0; kb < NH; ++kb)
                        hv[s2][kb] = sAct[l - 1][(4 * (NH * half + kb) + (j >> 2)) * 64 + (4 * s2 + g) * 4 + (j & 3)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
                    for (int kb = 0; kb < NH; ++kb) {
                        acc[0][kb] = mfma16(hv[s2][kb], dv[s2].x, acc[0][kb]);
                        acc[1][kb] = mfma16(hv[s2][kb], dv[s2].y, acc[1][kb]);
                    }
                    if (half == 1) {  // block NQ has A = 1: the bias gradient
                        acc[0][NH] = mfma16(1.0f, dv[s2].x, acc[0][NH]);
                        acc[1][NH] = mfma16(1.0f, dv[s2].y, acc[1][NH]);
                    }
                }
#pragma unroll
                for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                    for (int kb = 0; kb < NH; ++kb)
                        *reinterpret_cast<float4*>(gWl + (size_t)(32 * w + 2 * j + ub) * H + 16 * (NH * half + kb) + 4 * g) =
                            make_float4(acc[ub][kb][0], acc[ub][kb][1], acc[ub][kb][2], acc[ub][kb][3]);
                if (half == 1 && g == 0)
                    *reinterpret_cast<float2*>(gWl + H * H + 32 * w + 2 * j) = make_float2(acc[0][NH][0], acc[1][NH][0]);
            }
        }
        // ---- dH_{l-1} of the own units = W_l^T dZ_l, then through the ReLU / dropout mask of H_{l-1}
        {
            v4f16 d[2] = {v4f16{0.0f, 0.0f, 0.0f, 0.0f}, v4f16{0.0f, 0.0f, 0.0f, 0.0f}};
            product(wbuf[l - 1], sDz, d);
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                for (int r = 0; r < 4; ++r) dz[ub][r] = hreg[l - 1][ub][r] > 0.0f ? d[ub][r] * a.inv_keep : 0.0f;
        }
        __syncthreads();  // every wave is done with sDz
    }

    // ---- gW1 (own units x 8 inputs, bias in column 7): the same transposed contraction, MFMA rows = inputs
    put(sDz, dz);
    __syncthreads();
    {
        v4f16 acc[2] = {v4f16{0.0f, 0.0f, 0.0f, 0.0f}, v4f16{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const float2 dv = own_pair(sDz, s2);
            const float xv = j < 8 ? sX[j * 16 + 4 * s2 + g] : 0.0f;
            acc[0] = mfma16(xv, dv.x, acc[0]);
            acc[1] = mfma16(xv, dv.y, acc[1]);
        }
        if (g < 2) {  // registers = inputs 4 g .. 4 g + 3 of unit 32 w + 2 j + ub
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
                *reinterpret_cast<float4*>(out + (32 * w + 2 * j + ub) * 8 + 4 * g) =
                    make_float4(acc[ub][0], acc[ub][1], acc[ub][2], acc[ub][3]);
        }
    }
    if (w == 0) {
        float gbo = g == 0 ? dout : 0.0f, loss = g == 0 ? diff * diff : 0.0f;
#pragma unroll
        for (int mk = 1; mk < 16; mk <<= 1) {
            gbo += __shfl_xor(gbo, mk, 64);
            loss += __shfl_xor(loss, mk, 64);
        }
        if (lane == 0) {
            out[NP - 1] = gbo;
            out[NP] = loss;
        }
    }
}

template <int H, int L>
__global__ __launch_bounds__(H * 2) void mlp_train_q16_kernel(MlpQuadArgs a)
{
    mlp_train_q16_body<H, L>(a, blockIdx.x);
}

// ---- many small networks trained side by side (the curve entry points: one net per curve point) ----------------
// One table row per problem; blockIdx.y = problem, the step index within the epoch is a kernel argument.  A
// workgroup builds its problem's argument block exactly as quad_steps() does on the host (same float conversions:
// the IEEE double division / square root on the device round like the host's) and runs the single-problem body,
// so every network ends the epoch with the bits of its own omc_mlp_train_epoch call.  Problems whose epoch is
// shorter leave at once.
struct MlpBatchProb {
    const float* data;
    float* params;
    float* m;
    float* v;
    float* partial;
    float* wt;
    double* loss_acc;
    int64_t nrows, batch, first_step;
    double lr, beta1, beta2, eps, wd;
    Shuffle shuf;
    uint32_t keep16, k0, k1;
    float inv_keep;
    int pstride, q16_rows;  // q16_rows: minibatches of up to so many rows run in 16-row tiles (0: never)
    // non-null: the set size lives in device memory (the per-step ContNet flow: the regression set of the step,
    // counted by the kernels right before): nrows = batch = (int64_t)*nrows_dev, read when the kernel runs
    const double* nrows_dev;
};

__device__ __forceinline__ void batch_rows(const MlpBatchProb& p, int64_t* nrows, int64_t* batch)
{
    if (p.nrows_dev) {
        *nrows = *batch = (int64_t)*p.nrows_dev;
    } else {
        *nrows = p.nrows;
        *batch = p.batch;
    }
}

// Q16: this launch serves the problems whose minibatch runs in 16-row tiles (the kernel their single call runs,
// mlp_train_kernel_choice); the others leave at once -- and the other way round in the 32-row launch.
template <int H, int L, bool Q16 = false>
__global__ __launch_bounds__(H * 2) void mlp_train_quad_batch_kernel(const MlpBatchProb* __restrict__ tab, int s,
                                                                     int step_base)
{
    const MlpBatchProb& p = tab[blockIdx.y];
    const int tile = blockIdx.x;
    int64_t nrows, batch;
    batch_rows(p, &nrows, &batch);
    if ((batch <= (int64_t)p.q16_rows) != Q16) return;
    const int64_t o = (int64_t)s * batch;
    if (o >= nrows) return;
    const int64_t nb = (nrows - o < batch) ? nrows - o : batch;
    constexpr int kRows = Q16 ? 16 : 32;
    MlpQuadArgs a;
    a.data = p.data;
    a.params = p.params;
    a.wt = p.wt;
    a.partial = p.partial;
    a.row0 = o;
    a.nrows = nb;
    a.shuf = p.shuf;
    a.ntiles = (int)((nb + kRows - 1) / kRows);
    a.pstride = p.pstride;
    a.two_over_b = (float)(2.0 / (double)nb);
    a.keep16 = p.keep16;
    a.inv_keep = p.inv_keep;
    a.step = (uint32_t)(p.first_step + step_base + s + 1);
    a.k0 = p.k0;
    a.k1 = p.k1;
    if constexpr (Q16) mlp_train_q16_body<H, L>(a, tile);  // (T covers the largest problem's tiles)
    else mlp_train_quad_body<H, L>(a, tile);
}

// Work-list form of the same launch for batches whose problems differ wildly in size (the per-step ContNet flow of
// a curve: at any loop step a few problems are at their first regression step with thousands of rows while the rest
// have a few dozen): `prefix[p]` = tiles of problems 0 .. p-1 (mlp_tile_prefix_kernel, once per time step), the
// grid's workgroups share the total evenly, each walking a contiguous run of (problem, tile) items.  One partial per
// tile as before, so nothing changes for the sums.
__global__ __launch_bounds__(1024) void mlp_tile_prefix_kernel(const MlpBatchProb* __restrict__ tab, int n,
                                                              int* __restrict__ prefix)
{
    __shared__ int seg[1024];
    const int tid = threadIdx.x;
    const int len = (n + 1023) / 1024, lo = tid * len, hi = lo + len < n ? lo + len : n;
    int s = 0;
    for (int i = lo; i < hi; ++i) {
        int64_t nrows, batch;
        batch_rows(tab[i], &nrows, &batch);
        const int64_t nb = nrows < batch ? nrows : batch;
        s += (int)((nb + 31) / 32);
    }
    seg[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? seg[tid - d] : 0;
        __syncthreads();
        seg[tid] += v;
        __syncthreads();
    }
    int run = tid ? seg[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        prefix[i] = run;
        int64_t nrows, batch;
        batch_rows(tab[i], &nrows, &batch);
        const int64_t nb = nrows < batch ? nrows : batch;
        run += (int)((nb + 31) / 32);
    }
    if (tid == 1023) prefix[n] = seg[1023];
}

template <int H, int L>
__global__ __launch_bounds__(H * 2) void mlp_train_quad_list_kernel(const MlpBatchProb* __restrict__ tab,
                                                                    const int* __restrict__ prefix, int n, int step_base)
{
    const int total = prefix[n];
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    int item = (int)blockIdx.x * per;
    const int end = item + per < total ? item + per : total;
    if (item >= end) return;
    int lo = 0, hi = n;  // the problem that owns `item`: the last p with prefix[p] <= item
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= item) lo = mid; else hi = mid;
    }
    int p = lo;
    while (item < end) {
        while (prefix[p + 1] <= item) ++p;  // (problems without tiles are skipped)
        const MlpBatchProb& pb = tab[p];
        int64_t nrows, batch;
        batch_rows(pb, &nrows, &batch);
        const int64_t nb = nrows < batch ? nrows : batch;
        MlpQuadArgs a;
        a.data = pb.data;
        a.params = pb.params;
        a.wt = pb.wt;
        a.partial = pb.partial;
        a.row0 = 0;
        a.nrows = nb;
        a.shuf = pb.shuf;
        a.ntiles = (int)((nb + 31) / 32);
        a.pstride = pb.pstride;
        a.two_over_b = (float)(2.0 / (double)nb);
        a.keep16 = pb.keep16;
        a.inv_keep = pb.inv_keep;
        a.step = (uint32_t)(pb.first_step + step_base + 1);
        a.k0 = pb.k0;
        a.k1 = pb.k1;
        const int first = prefix[p];
        const int last = prefix[p + 1] < end ? prefix[p + 1] : end;
        for (; item < last; ++item) {
            mlp_train_quad_body<H, L>(a, item - first);
            __syncthreads();
        }
    }
}

// bc1 / bc2: 1 - beta^step for step = 0 .. (host-computed tables: libm pow, as quad_steps uses)
template <bool FLAT>
__global__ __launch_bounds__(256) void mlp_adam_batch_kernel(const MlpBatchProb* __restrict__ tab, int s, int step_base,
                                                            int H, int L, const double* __restrict__ bc1,
                                                            const double* __restrict__ bc2)
{
    const MlpBatchProb& p = tab[blockIdx.y];
    int64_t nrows, batch;
    batch_rows(p, &nrows, &batch);
    const int64_t o = (int64_t)s * batch;
    if (o >= nrows) return;
    const int64_t nb = (nrows - o < batch) ? nrows - o : batch;
    const int64_t step = p.first_step + step_base + s + 1;
    MlpAdamArgs b;
    b.params = p.params;
    b.m = p.m;
    b.v = p.v;
    b.partial = p.partial;
    b.loss_acc = p.loss_acc;
    b.nparts = batch <= (int64_t)p.q16_rows ? (int)((nb + 15) / 16) : (int)((nb + 31) / 32);  // one partial per tile
    b.nparams = mlp_params_of(H, L);
    b.stride = p.pstride;
    b.wt = p.wt; b.H = H; b.L = L;
    b.inv_b = (float)(1.0 / (double)nb);
    b.lr_t = (float)(p.lr / bc1[step]);
    b.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2[step]));
    b.beta1 = (float)p.beta1;
    b.beta2 = (float)p.beta2;
    b.eps = (float)p.eps;
    b.wd = (float)p.wd;
    // FLAT: one thread per parameter (few, full workgroups: many problems per launch); else 16 threads per parameter
    // (the single-problem kernel's shape: shortest latency for a lone problem).  Same bits either way.
    if constexpr (FLAT) mlp_adam_body_flat(b, (int)(blockIdx.x * 256 + threadIdx.x));
    else mlp_adam_body(b);
}

__global__ __launch_bounds__(256) void mlp_transpose_batch_kernel(const MlpBatchProb* __restrict__ tab, int H, int L)
{
    const MlpBatchProb& p = tab[blockIdx.y];
    const int n = (L - 1) * H * H;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int j = i / (H * H), rem = i - j * H * H, k = rem / H, u = rem - k * H;
        p.wt[i] = p.params[H * 8 + (size_t)j * (H * H + H) + (size_t)u * H + k];
    }
}

// wt_j[k][i] = W_j[i][k] for the L-1 connections (start of an epoch; Adam keeps it current)
__global__ __launch_bounds__(256) void mlp_transpose_kernel(const float* params, float* wt, int H, int L)
{
    const int n = (L - 1) * H * H;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int j = idx / (H * H), rem = idx - j * H * H, k = rem / H, i = rem - k * H;
        wt[idx] = params[H * 8 + j * (H * H + H) + i * H + k];
    }
}

template <int H, int L>
__global__ __launch_bounds__(256) void mlp_apply_kernel(MlpApplyArgs a)
{
    constexpr int NT = H / 32, LDW = H + 1;
    extern __shared__ float sw[];
    float* sW1 = sw;                       // [H][9]
    float* sWh = sW1 + H * kLdW1;          // (L-1) x { [H][H+1], bias [H] }
    float* sWo = sWh + (L - 1) * (H * LDW + H);
    float* sBo = sWo + H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    // (stage_block: a chunk's loads are all in flight before its first LDS store, omc_device.h)
    stage_block(a.params, H * 8, tid, [&](int i, float v) { sW1[(i >> 3) * kLdW1 + (i & 7)] = v; });
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const float* src = a.params + H * 8 + l * (H * H + H);
        float* dst = sWh + l * (H * LDW + H);
        stage_block(src, H * H + H, tid, [&](int i, float v) {  // weights [H][H] -> [H][H + 1], then the bias row
            if (i < H * H) dst[(i / H) * LDW + (i % H)] = v;
            else dst[H * LDW + (i - H * H)] = v;
        });
    }
    stage_block(a.params + H * 8 + (L - 1) * (H * H + H), H + 1, tid, [&](int i, float v) {
        if (i < H) sWo[i] = v;
        else sBo[0] = v;
    });
    __syncthreads();
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.ntiles) return;  // whole wave; no barrier below
    const int64_t p = (int64_t)tile * 32 + c;
    const bool live = p < a.M;
    const float* col = a.S + (live ? p : a.M - 1);
    const uint32_t pk = (uint32_t)(p < a.key_half ? p + a.key_base0 : p - a.key_half + a.key_base1);  // dropout key
    const double K = a.K;
    const uint32_t rtag = (uint32_t)h + 2u * (uint32_t)(p >> 32);
    float sx = col[(int64_t)a.N * a.ld];
    int tex = a.N;
    bool done = !live;
    float s_next = a.N > 1 ? col[(int64_t)(a.N - 1) * a.ld] : 0.0f;
    for (int t = a.N - 1; t >= 1; --t) {
        const float sf = s_next;
        if (t > 1) s_next = col[(int64_t)(t - 1) * a.ld];
        const double sd = (double)sf;
        const double imm = a.is_put ? K - sd : sd - K;
        const bool need = !done && imm > 0.0;
        if (__builtin_amdgcn_ballot_w64(need) == 0) continue;  // nobody to decide for (uniform)
        const double x = sd / K;
        const double st = sqrt(fmax(a.T - (double)t * a.dt, 1e-6));
        float4 xin;
        if (h == 0) {
            xin.x = (float)((1.0 - a.fm[0]) * a.rs[0]);
            xin.y = (float)((x - a.fm[1]) * a.rs[1]);
            xin.z = (float)((x * x - a.fm[2]) * a.rs[2]);
            xin.w = (float)((x * x * x - a.fm[3]) * a.rs[3]);
        } else {
            xin.x = (float)((fmax(x - 1.0, 0.0) - a.fm[4]) * a.rs[4]);
            xin.y = (float)((st - a.fm[5]) * a.rs[5]);
            xin.z = (float)((x * st - a.fm[6]) * a.rs[6]);
            xin.w = 1.0f;  // bias input
        }
        v16f act[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) act[mt][r] = 0.0f;
            const float* wr = sW1 + (32 * mt + c) * kLdW1 + 4 * h;
            act[mt] = mfma(wr[0], xin.x, act[mt]);
            act[mt] = mfma(wr[1], xin.y, act[mt]);
            act[mt] = mfma(wr[2], xin.z, act[mt]);
            act[mt] = mfma(wr[3], xin.w, act[mt]);
        }
        relu_dropout_n<NT>(act, pk, (uint32_t)t, 0x300u + rtag, a.keep16, a.inv_keep, a.k0, a.k1);
#pragma unroll
        for (int l = 0; l < L - 1; ++l) {
            const float* W = sWh + l * (H * LDW + H);
            const float* B = W + H * LDW;
            v16f nxt[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) nxt[mt][r] = B[unit_of(mt, r, h)];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int k = unit_of(kt, s, h);
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt) nxt[mt] = mfma(W[(32 * mt + c) * LDW + k], act[kt][s], nxt[mt]);
                }
            }
            relu_dropout_n<NT>(nxt, pk, (uint32_t)t, 0x400u + 0x100u * (uint32_t)l + rtag, a.keep16,
                               a.inv_keep, a.k0, a.k1);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) act[mt] = nxt[mt];
        }
        float o = 0.0f;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o = __builtin_fmaf(sWo[unit_of(mt, r, h)], act[mt][r], o);
        o += __shfl_xor(o, 32, 64);
        o += sBo[0];
        const double cont = (double)o * a.ysd + a.ym;
        if (need && imm > cont) {
            done = true;
            tex = t;
            sx = sf;
        }
    }
    if (live && h == 0) {
        a.sx[p] = sx;
        a.tex[p] = tex;
    }
}

// ------------------------------------------------------------------ local-vol paths (row f-4)
// simulate_local_vol_paths_antithetic (options_model_3.py:300-333) with the implied-vol network
// (ImprovedIVNetwork, NN_training_stock_iv.py:109-155: Linear(2,64)+GELU, L x [h += GELU(
// LayerNorm(Linear(h)))], Linear(64,1) clamped at epsilon; dropout is off in eval mode) evaluated
// inside the path loop: one wave carries 32 columns through all time steps, activations stay in
// the transposed MFMA accumulator layout of the trainer above (lane <-> column), so LayerNorm's
// sums over the 64 units are sums over a lane's registers plus one swap between half-waves.
// Flat parameters: Win|bin as [64][4] (w_m, w_tau, bias, 0), per layer W [64][64], b, gamma,
// beta [64] each, then the output weights [64] and bias [1].
struct LocalVolArgs {
    float* S;
    int64_t ld, M, P;
    int N, L;
    const float* params;
    const float* Z;  // [N][P] normals of the first half; the partner column uses -z
    float s0, r, dt, sqdt, eps_out;
    double K, T, dtd, inv_m_scale, inv_tau_scale;
    int ntiles;
};

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__global__ __launch_bounds__(256) void localvol_paths_kernel(LocalVolArgs a)
{
    constexpr int kLayer = kH * kLdW2 + 3 * kH;
    extern __shared__ float sw[];
    float* sWin = sw;                 // [64][4]
    float* sLay = sWin + kH * 4;      // L x { W [64][65], b, gamma, beta }
    float* sWo = sLay + a.L * kLayer;  // [64] + bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    stage_block(a.params, kH * 4, tid, [&](int i, float v) { sWin[i] = v; });
    for (int l = 0; l < a.L; ++l) {
        const float* src = a.params + kH * 4 + l * (kH * kH + 3 * kH);
        float* dst = sLay + l * kLayer;
        stage_block(src, kH * kH + 3 * kH, tid, [&](int i, float v) {
            if (i < kH * kH) dst[(i >> 6) * kLdW2 + (i & 63)] = v;
            else dst[kH * kLdW2 + (i - kH * kH)] = v;
        });
    }
    stage_block(a.params + kH * 4 + a.L * (kH * kH + 3 * kH), kH + 1, tid, [&](int i, float v) { sWo[i] = v; });
    __syncthreads();
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.ntiles) return;  // whole wave; no barrier below
    const int64_t col = (int64_t)tile * 32 + c;
    const bool live = col < a.M;
    const int64_t zc = live ? (col < a.P ? col : col - a.P) : 0;
    const float zs = col < a.P ? 1.0f : -1.0f;
    float s = a.s0;
    if (live && h == 0) a.S[col] = s;
    float z_next = a.Z[zc];
    for (int t = 1; t <= a.N; ++t) {
        const float z = z_next * zs;
        if (t < a.N) z_next = a.Z[(int64_t)t * a.P + zc];
        const double tau = fmax(a.T - (double)(t - 1) * a.dtd, 1e-6);
        const float xin = h == 0 ? (float)(log(fmax(a.K, 1e-8) / fmax((double)s, 1e-8)) * a.inv_m_scale)
                                 : (float)(tau * a.inv_tau_scale);
        v16f act[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) act[mt][r] = sWin[unit_of(mt, r, h) * 4 + 2];
            act[mt] = mfma(sWin[(32 * mt + c) * 4 + h], xin, act[mt]);
#pragma unroll
            for (int r = 0; r < 16; ++r) act[mt][r] = gelu_exact(act[mt][r]);
        }
        for (int l = 0; l < a.L; ++l) {
            const float* W = sLay + l * kLayer;
            const float* B = W + kH * kLdW2;
            v16f zz[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) zz[mt][r] = B[unit_of(mt, r, h)];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int sI = 0; sI < 16; ++sI) {
                    const int k = unit_of(kt, sI, h);
                    zz[0] = mfma(W[(c)*kLdW2 + k], act[kt][sI], zz[0]);
                    zz[1] = mfma(W[(32 + c) * kLdW2 + k], act[kt][sI], zz[1]);
                }
            }
            // LayerNorm over the 64 units of a column: this lane's 32 + the other half-wave's 32
            float sum = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += zz[mt][r];
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / 64.0f);
            float sq = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = zz[mt][r] - mean;
                    sq = __builtin_fmaf(d, d, sq);
                }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = 1.0f / __builtin_sqrtf(sq * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = unit_of(mt, r, h);
                    const float y = (zz[mt][r] - mean) * rstd * B[kH + u] + B[2 * kH + u];
                    act[mt][r] += gelu_exact(y);
                }
        }
        float o = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o = __builtin_fmaf(sWo[unit_of(mt, r, h)], act[mt][r], o);
        o += __shfl_xor(o, 32, 64);
        o += sWo[kH];
        const float sig = fmaxf(fmaxf(o, a.eps_out), 1e-6f);
        s = s * expf((a.r - 0.5f * sig * sig) * a.dt + sig * a.sqdt * z);
        if (live && h == 0) a.S[(int64_t)t * a.ld + col] = s;
    }
}

}  // namespace

size_t nn_stats_scratch_bytes() { return sizeof(double) * 8 * (1024 + 2); }

// out[0..7] = means, out[8..15] = population variances (slots 0-5 features 1..6, slot 6 target)
hipError_t nn_feature_stats(hipStream_t st, const double* x, const int32_t* t, const double* y, int64_t n,
                            double T, double dt, double* scratch, double* out16)
{
    StatArgs a;
    a.x = x; a.t = t; a.y = y; a.n = n; a.T = T; a.dt = dt;
    a.part = scratch; a.pstride = 1024; a.mean = out16;
    int nblk = (int)((n + kBlock * 8 - 1) / (kBlock * 8));
    nblk = nblk < 1 ? 1 : (nblk > 1024 ? 1024 : nblk);
    hipLaunchKernelGGL(nn_stats_kernel<0>, dim3(nblk), dim3(kBlock), 0, st, a);
    hipLaunchKernelGGL(nn_stats_finish_kernel, dim3(1), dim3(kBlock), 0, st, scratch, nblk, 1024, (double)n, out16);
    hipLaunchKernelGGL(nn_stats_kernel<1>, dim3(nblk), dim3(kBlock), 0, st, a);
    hipLaunchKernelGGL(nn_stats_finish_kernel, dim3(1), dim3(kBlock), 0, st, scratch, nblk, 1024, (double)n, out16 + 8);
    return hipGetLastError();
}

namespace {

Shuffle make_shuffle(int64_t n, uint64_t key)
{
    Shuffle sh;
    sh.n = (uint64_t)n;
    sh.on = (key != 0 && n > 1) ? 1 : 0;
    uint32_t bits = 2;
    while (bits < 62 && (1ull << bits) < sh.n) ++bits;
    sh.abits = bits / 2;
    sh.bbits = bits - sh.abits;
    for (int i = 0; i < 4; ++i) {  // splitmix64 of the key -> round keys
        uint64_t z = key + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        sh.key[i] = (uint32_t)((z ^ (z >> 31)) >> 16);
    }
    return sh;
}

__global__ __launch_bounds__(256) void mlp_shuffle_kernel(Shuffle s, int64_t* out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < s.n) out[i] = (int64_t)shuffle_index(s, i);
}

// ---- sharded epochs: this rank's positions of the epoch's keyed permutation over ALL ranks' rows
struct ShardSel {
    Shuffle sh;             // permutation of [0, rows_global)
    const int64_t* gstart;  // [nseg + 1]
    const int64_t* lstart;  // [nseg], -1 = not this rank's
    int nseg;
    int group;              // segments per time step (2 x ranks): the table is searched in two levels -- the step in LDS
                            // (nseg / group + 1 starts), the segment inside the step's `group` entries; 0: flat search
};
constexpr int kSelPerThread = 16, kSelPerBlock = 256 * kSelPerThread;
constexpr int kSelMaxSteps = 4096;  // step starts staged in LDS (32 KB); longer tables are searched flat

// own row of global row g, or -1: the LAST segment that starts at or before g (empty segments share a start)
__device__ __forceinline__ int64_t shard_locate(const ShardSel& s, const int64_t* __restrict__ step_start, int nsteps,
                                                int64_t g)
{
    int lo = 0, hi = s.nseg;  // gstart[lo] <= g < gstart[hi]
    if (step_start) {
        int a = 0, b = nsteps;  // step_start[a] <= g < step_start[b]
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (step_start[mid] <= g) a = mid;
            else b = mid;
        }
        lo = a * s.group;
        hi = min(lo + s.group, s.nseg);
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s.gstart[mid] <= g) lo = mid;
        else hi = mid;
    }
    const int64_t l = s.lstart[lo];
    return l < 0 ? -1 : l + (g - s.gstart[lo]);
}

// Every rank walks ALL epoch positions (ownership is only known after the permutation is evaluated), so this is the one
// per-rank cost of a sharded epoch that does not shrink with the number of ranks: 16 positions per thread, the
// time-step starts of the segment table in LDS.  WRITE = false: which of a thread's 16 positions are this rank's ->
// mask[thread] (a bit each), cnt[block] = their number.  WRITE = true: only the positions with a bit set are evaluated
// again (1 / ranks of them) and written, ascending, at offs[block] + rank inside the block.
template <bool WRITE>
__global__ __launch_bounds__(256) void shard_select_kernel(ShardSel s, int32_t* __restrict__ cnt, uint16_t* __restrict__ mask,
                                                           const int64_t* __restrict__ offs, int64_t* __restrict__ sel_row,
                                                           int64_t* __restrict__ sel_i)
{
    __shared__ int scan[256];
    __shared__ int64_t sh_step[kSelMaxSteps + 1];
    const int tid = threadIdx.x;
    const int nsteps = s.group > 0 ? s.nseg / s.group : 0;
    const int64_t* step_start = nullptr;
    if (nsteps > 0 && nsteps <= kSelMaxSteps && nsteps * s.group == s.nseg) {
        for (int i = tid; i <= nsteps; i += 256) sh_step[i] = s.gstart[min(i * s.group, s.nseg)];
        __syncthreads();
        step_start = sh_step;
    }
    const uint64_t i0 = (uint64_t)blockIdx.x * kSelPerBlock + (uint64_t)tid * kSelPerThread;
    const size_t slot = (size_t)blockIdx.x * 256 + tid;
    if (!WRITE) {
        unsigned m = 0;
#pragma unroll 4
        for (int j = 0; j < kSelPerThread; ++j) {
            const uint64_t i = i0 + j;
            if (i < s.sh.n && shard_locate(s, step_start, nsteps, (int64_t)shuffle_index(s.sh, i)) >= 0) m |= 1u << j;
        }
        mask[slot] = (uint16_t)m;
        int n = __builtin_popcount(m);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
        if ((tid & 63) == 0) scan[tid >> 6] = n;
        __syncthreads();
        if (tid == 0) cnt[blockIdx.x] = scan[0] + scan[1] + scan[2] + scan[3];
        return;
    }
    const unsigned m = mask[slot];
    const int n = __builtin_popcount(m);
    scan[tid] = n;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan of the threads' counts
        const int v = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int64_t o = offs[blockIdx.x] + (scan[tid] - n);
    for (unsigned r = m; r; r &= r - 1) {
        const int j = __builtin_ctz(r);
        const uint64_t i = i0 + j;
        sel_row[o] = shard_locate(s, step_start, nsteps, (int64_t)shuffle_index(s.sh, i));
        sel_i[o] = (int64_t)i;
        ++o;
    }
}

__global__ __launch_bounds__(256) void shard_gather_kernel(const float4* __restrict__ data, const int64_t* __restrict__ sel_row,
                                                           const int64_t* __restrict__ sel_i, int64_t n, int64_t batch,
                                                           float4* __restrict__ out, uint32_t* __restrict__ drop_pos)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per half row (16 bytes)
    const int64_t j = t >> 1;
    if (j >= n) return;
    const int h = (int)(t & 1);
    out[j * 2 + h] = data[sel_row[j] * 2 + h];
    if (h == 0) drop_pos[j] = (uint32_t)(sel_i[j] % batch);
}

// step_off[k] = first j with sel_i[j] >= k * batch (sel_i ascending), k = 0 .. steps
__global__ __launch_bounds__(256) void shard_step_off_kernel(const int64_t* __restrict__ sel_i, int64_t n, int64_t batch,
                                                             int64_t steps, int64_t* __restrict__ step_off)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > steps) return;
    const int64_t key = k * batch;
    int64_t lo = 0, hi = n;  // first index with sel_i >= key
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sel_i[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    step_off[k] = lo;
}

}  // namespace

// Which kernel trains (hidden, layers) at this minibatch size: 1 = workgroup kernel (64 units,
// weights in LDS; large batches), 2 = tile-per-wave kernel (64 units at small batches, 128 units at
// any batch), 0 = neither.
// 1: workgroup kernel (64 units, large batches); 2: tile-per-wave kernel; 3: tile-per-workgroup kernel
// (minibatches of at most kMlpMaxGroups tiles: the reference's batch of 256 rows is 8)
// the kernel before the 16-row tiles existed -- what the side-by-side (batched) trainers still run
static int mlp_train_kernel_choice32(int hidden, int layers, int64_t batch)
{
    if (layers != 2 && layers != 3) return 0;
    const int64_t tiles = (batch + 31) / 32;
    static const int quad = getenv("OMC_MLP_QUAD") ? atoi(getenv("OMC_MLP_QUAD")) : 1;
    if (hidden == 64) return tiles <= 32 ? (quad ? 3 : 2) : 1;
    if (hidden == 128) return (quad && tiles <= kMlpMaxGroups) ? 3 : 2;
    if (hidden == 32) return 3;  // 32 units (the per-step ContNet of omc_contnet.hip; SingleLSMNet(7, 32, 2 | 3)): any number of tiles
    return 0;
}

// 16-row tiles while a minibatch does not fill the chip (kMlpQ16MaxRows rows = 256 workgroups; measured at 3 x 128: 28.4
// against 30.2 us per step at 1,024 rows, a tie at 2,048, 37.4 against 40.0 at 4,096, 57 against 45 at 8,192): the reference's
// own min(256, R).  OMC_MLP_Q16=0 switches them off, =N moves the limit to N rows (A/B measurements).
int64_t mlp_q16_rows(int hidden)
{
    static const int q16 = getenv("OMC_MLP_Q16") ? atoi(getenv("OMC_MLP_Q16")) : -1;
    if (hidden != 64 && hidden != 128) return 0;
    return q16 < 0 ? kMlpQ16MaxRows : q16;
}

int mlp_train_kernel_choice(int hidden, int layers, int64_t batch)
{
    const int c = mlp_train_kernel_choice32(hidden, layers, batch);
    if (c == 3 && batch <= mlp_q16_rows(hidden)) return 4;
    return c;
}

int mlp_train_param_count(int hidden, int layers)
{
    if ((hidden != 32 && hidden != 64 && hidden != 128) || (layers != 2 && layers != 3)) return -1;
    return mlp_params_of(hidden, layers);
}

size_t mlp_partial_bytes(int hidden, int layers, int64_t batch)
{
    const int choice = mlp_train_kernel_choice(hidden, layers, batch);
    if (choice == 1) return sizeof(float) * (size_t)kMlpMaxGroups * kMlpPartialStride3;
    if (choice == 4) return sizeof(float) * (size_t)((batch + 15) / 16) * tile_pstride(hidden, layers);
    if (choice == 2 || choice == 3) {
        const int64_t tiles = (batch + 31) / 32, cap = tile_waves_max(hidden);
        // 32 units: one partial per tile however many (the kernel never accumulates across tiles)
        return sizeof(float) * (size_t)(hidden == 32 || tiles < cap ? tiles : cap) * tile_pstride(hidden, layers);
    }
    return 0;
}

size_t mlp_wt_bytes(int hidden, int layers) { return sizeof(float) * (size_t)(layers - 1) * hidden * hidden; }

// ---- which units does dropout keep?  (omc_mlp_dropout_masks: the device half of the mask oracle's known-answer test)
// Runs the very device functions the trainers and pass 2 call -- relu_dropout / _t / _n / _1 -- on activations of 1.0
// with each kernel's tags and its register -> hidden-unit map, and writes keep / drop per (layer, row, unit).
// variant 0: mlp_apply_kernel (key = path column, step = time step), 1: mlp_train_kernel, 2: mlp_train_tile_kernel,
// 3: mlp_train_quad_kernel (key = position in the minibatch, step = optimizer step).
namespace {
template <int H>
__global__ __launch_bounds__(64) void mlp_mask_probe_kernel(int variant, int layers, int64_t n_rows, const uint32_t* keys,
                                                            uint32_t step, uint32_t keep16, float inv_keep, uint32_t k0,
                                                            uint32_t k1, uint8_t* __restrict__ out)
{
    constexpr int NT = H / 32;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * 32 + c;
    if (row >= n_rows) return;
    const uint32_t key = keys ? keys[row] : (uint32_t)row;
    auto rho = [&](int r) { return (r >> 2) * 8 + 4 * h + (r & 3); };
    for (int j = 0; j < layers; ++j) {
        uint8_t* o = out + ((size_t)j * (size_t)n_rows + (size_t)row) * H;
        v16f z[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[mt][r] = 1.0f;
        if (variant == 4) {  // 16-row tiles: lane (j, g) of wave w holds units 32 w + 8 g + 2 r + ub; here c <-> j, 16 rows per half-wave h
#pragma unroll
            for (int oct = 0; oct < H / 8; ++oct) {
                v4f16 zz[2] = {v4f16{1.0f, 1.0f, 1.0f, 1.0f}, v4f16{1.0f, 1.0f, 1.0f, 1.0f}};
                relu_dropout_q16(zz, key, step, 0x100u * (uint32_t)(j + 1) + (uint32_t)oct, keep16, inv_keep, k0, k1);
                if (h == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[8 * oct + 2 * r] = zz[0][r] != 0.0f;
                        o[8 * oct + 2 * r + 1] = zz[1][r] != 0.0f;
                    }
                }
            }
            continue;
        }
        if (variant == 3) {
#pragma unroll
            for (int w = 0; w < NT; ++w) {
                relu_dropout_1(z[w], key, step, 0x100u * (uint32_t)(j + 1) + 0x10u * (uint32_t)w + (uint32_t)h, keep16,
                               inv_keep, k0, k1);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[32 * w + rho(r)] = z[w][r] != 0.0f;
            }
            continue;
        }
        if constexpr (NT == 1) {
            if (variant == 0) {  // pass 2 with one 32-unit tile
                const uint32_t tag = (j == 0 ? 0x300u : 0x400u + 0x100u * (uint32_t)(j - 1)) + (uint32_t)h;
                relu_dropout_n<1>(z, key, step, tag, keep16, inv_keep, k0, k1);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[unit_of(0, r, h)] = z[0][r] != 0.0f;
            }
        }
        if constexpr (NT >= 2) {
            if (variant == 2) {
                relu_dropout_t<NT, true>(z, key, step, 0x100u * (uint32_t)(j + 1) + (uint32_t)h, keep16, inv_keep, k0, k1);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[NT * rho(r) + mt] = z[mt][r] != 0.0f;
            } else if (variant == 1) {
                if constexpr (NT == 2) {
                    relu_dropout<false>(z, key, step, 0x100u * (uint32_t)(j + 1) + (uint32_t)h, keep16, inv_keep, k0, k1);
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[unit_of(mt, r, h)] = z[mt][r] != 0.0f;
                }
            } else {
                const uint32_t tag = (j == 0 ? 0x300u : 0x400u + 0x100u * (uint32_t)(j - 1)) + (uint32_t)h;
                relu_dropout_n<NT>(z, key, step, tag, keep16, inv_keep, k0, k1);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[unit_of(mt, r, h)] = z[mt][r] != 0.0f;
            }
        }
    }
}
}  // namespace

hipError_t mlp_dropout_masks(hipStream_t st, int variant, int hidden, int layers, int64_t n_rows, const uint32_t* keys,
                             uint32_t step, uint64_t seed, double dropout, uint8_t* out)
{
    const uint32_t keep16 = dropout > 0.0 ? (uint32_t)llround((1.0 - dropout) * 65536.0) : 65536u;
    const float inv_keep = keep16 >= 65536u ? 1.0f : (float)(65536.0 / (double)keep16);
    const dim3 grid((unsigned)((n_rows + 31) / 32)), block(64);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (hidden == 32) hipLaunchKernelGGL(mlp_mask_probe_kernel<32>, grid, block, 0, st, variant, layers, n_rows, keys, step, keep16, inv_keep, k0, k1, out);
    else if (hidden == 64) hipLaunchKernelGGL(mlp_mask_probe_kernel<64>, grid, block, 0, st, variant, layers, n_rows, keys, step, keep16, inv_keep, k0, k1, out);
    else if (hidden == 128) hipLaunchKernelGGL(mlp_mask_probe_kernel<128>, grid, block, 0, st, variant, layers, n_rows, keys, step, keep16, inv_keep, k0, k1, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}


hipError_t mlp_shuffle_indices(hipStream_t st, int64_t n, uint64_t shuffle_key, int64_t* out)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mlp_shuffle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       make_shuffle(n, shuffle_key), out);
    return hipGetLastError();
}

size_t mlp_shard_scratch_bytes(int64_t rows_global)
{
    const size_t nb = (size_t)((rows_global + kSelPerBlock - 1) / kSelPerBlock);
    return sizeof(int64_t) * (nb + 2) + sizeof(int32_t) * (nb + 2) + sizeof(uint16_t) * 256 * (nb + 1) + 64;
}

hipError_t mlp_shard_select(hipStream_t st, int64_t rows_global, uint64_t shuffle_key, const int64_t* gstart,
                            const int64_t* lstart, int nseg, int group, void* scratch, int64_t* sel_row, int64_t* sel_i,
                            const int64_t** total_dev)
{
    const int64_t nb = (rows_global + kSelPerBlock - 1) / kSelPerBlock;
    int64_t* offs = (int64_t*)scratch;
    int32_t* cnt = (int32_t*)(offs + nb + 2);
    uint16_t* mask = (uint16_t*)(((uintptr_t)(cnt + nb + 2) + 15) & ~(uintptr_t)15);
    *total_dev = offs + nb;
    if (nb == 0 || nseg <= 0) return hipMemsetAsync(offs, 0, sizeof(int64_t) * (size_t)(nb + 1), st);
    ShardSel s;
    s.sh = make_shuffle(rows_global, shuffle_key);
    s.gstart = gstart; s.lstart = lstart; s.nseg = nseg; s.group = group;
    hipLaunchKernelGGL(shard_select_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, s, cnt, mask, (const int64_t*)nullptr,
                       (int64_t*)nullptr, (int64_t*)nullptr);
    hipError_t e = nn_scan_counts(st, cnt, nb, offs);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(shard_select_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, s, cnt, mask, (const int64_t*)offs,
                       sel_row, sel_i);
    return hipGetLastError();
}

hipError_t mlp_shard_gather(hipStream_t st, const float* data, const int64_t* sel_row, const int64_t* sel_i,
                            int64_t n_local, int64_t batch, int64_t steps, float* data_epoch, uint32_t* drop_pos,
                            int64_t* step_off)
{
    if (n_local > 0)
        hipLaunchKernelGGL(shard_gather_kernel, dim3((unsigned)((2 * n_local + 255) / 256)), dim3(256), 0, st,
                           (const float4*)data, sel_row, sel_i, n_local, batch, (float4*)data_epoch, drop_pos);
    hipLaunchKernelGGL(shard_step_off_kernel, dim3((unsigned)((steps + 1 + 255) / 256)), dim3(256), 0, st, sel_i, n_local, batch,
                       steps, step_off);
    return hipGetLastError();
}

// The optimizer steps of one epoch: step k trains on `local` rows starting at `row0` of t.data and is one minibatch of
// `global` rows (single GPU: the same thing; sharded: this rank's part of the job's minibatch).
struct StepSpan { int64_t row0, local, global; };
static inline int64_t plan_steps(const MlpTrainPlan& t)
{
    const int64_t rows = t.step_off ? t.rows_global : t.nrows;
    return (rows + t.batch - 1) / t.batch;
}
static inline StepSpan plan_span(const MlpTrainPlan& t, int64_t k)
{
    const int64_t rows = t.step_off ? t.rows_global : t.nrows;
    const int64_t o = k * t.batch, g = rows - o < t.batch ? rows - o : t.batch;
    if (!t.step_off) return {o, g, g};
    return {t.step_off[k], t.step_off[k + 1] - t.step_off[k], g};
}
static inline Shuffle plan_shuffle(const MlpTrainPlan& t)
{
    return t.step_off ? make_shuffle(t.nrows, 0) : make_shuffle(t.nrows, t.shuffle_key);  // sharded: rows are in epoch order
}
// After the forward / backward launch(es) of a step: Adam on this rank's partials, or -- sharded -- on the sum of all
// ranks' partials.  `b` is complete except for the partial source.  -> 0 or the all-reduce's error.
static inline void launch_adam(hipStream_t st, const MlpAdamArgs& b, int adam_blocks)
{
    hipLaunchKernelGGL(mlp_adam_kernel, dim3(adam_blocks), dim3(256), 0, st, b);
}

static int finish_step(hipStream_t st, const MlpTrainPlan& t, MlpAdamArgs b, int adam_blocks)
{
    if (!t.step_off) {
        launch_adam(st, b, adam_blocks);
        return 0;
    }
    const int n = b.nparams + 1;
    hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b.partial, b.nparts, b.stride,
                       b.nparams, t.gred);
    const int rc = t.allreduce(t.allreduce_user, t.gred, n);
    if (rc) return rc;
    float* gf = reinterpret_cast<float*>(t.gred + n);  // [n] floats behind the doubles
    hipLaunchKernelGGL(mlp_grad_to_float_kernel, dim3((n + 255) / 256), dim3(256), 0, st, t.gred, n, gf);
    b.partial = gf;
    b.nparts = 1;
    b.stride = n;
    hipLaunchKernelGGL(mlp_adam_kernel, dim3(adam_blocks), dim3(256), 0, st, b);
    return 0;
}

template <int L>
static hipError_t train_steps(hipStream_t st, const MlpTrainPlan& t)
{
    static std::atomic<uint64_t> attr_mask{0};  // per device (omc_kernels.h)
    const size_t lds_bytes = sizeof(float) * (size_t)train_lds_floats(L);
    if (true) {
        hipError_t e = set_max_dynamic_lds(attr_mask, reinterpret_cast<const void*>(mlp_train_kernel<L>), lds_bytes);
        if (e != hipSuccess) return e;
    }
    const Shuffle sh = plan_shuffle(t);
    int64_t step = t.first_step;
    const int64_t nsteps = plan_steps(t);
    for (int64_t k = 0; k < nsteps; ++k) {
        const StepSpan sp = plan_span(t, k);
        const int64_t o = sp.row0, nb = sp.local;
        ++step;
        MlpTrainArgs a;
        a.data = t.data;
        a.params = t.params;
        a.partial = t.partial;
        a.row0 = o;
        a.nrows = nb;
        a.shuf = sh;
        a.ntiles = (int)((nb + 31) / 32);
        a.two_over_b = (float)(2.0 / (double)sp.global);
        a.drop_pos = t.drop_pos ? t.drop_pos + o : nullptr;
        a.keep16 = t.dropout > 0.0 ? (uint32_t)llround((1.0 - t.dropout) * 65536.0) : 65536u;
        a.inv_keep = a.keep16 >= 65536u ? 1.0f : (float)(65536.0 / (double)a.keep16);
        a.step = (uint32_t)step;
        a.k0 = (uint32_t)t.seed;
        a.k1 = (uint32_t)(t.seed >> 32);
        int groups = (a.ntiles + 3) / 4;
        if (groups > kMlpMaxGroups) groups = kMlpMaxGroups;
        if (groups > 0) hipLaunchKernelGGL(mlp_train_kernel<L>, dim3(groups), dim3(256), lds_bytes, st, a);  // (a rank may own no row of a step)
        MlpAdamArgs b;
        b.params = t.params;
        b.m = t.adam_m;
        b.v = t.adam_v;
        b.partial = t.partial;
        b.loss_acc = t.loss_acc;
        b.nparts = groups;
        b.nparams = train_params(L);
        b.stride = g_stride(L);
        b.wt = nullptr; b.H = kH; b.L = L;
        b.inv_b = (float)(1.0 / (double)sp.global);
        const double bc1 = 1.0 - pow(t.beta1, (double)step), bc2 = 1.0 - pow(t.beta2, (double)step);
        b.lr_t = (float)(t.lr / bc1);
        b.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        b.beta1 = (float)t.beta1;
        b.beta2 = (float)t.beta2;
        b.eps = (float)t.eps;
        b.wd = (float)t.weight_decay;
        if (finish_step(st, t, b, (train_params(L) + 16) / 16)) return hipErrorUnknown;
    }
    return hipGetLastError();
}

template <int H, int L>
static hipError_t tile_steps(hipStream_t st, const MlpTrainPlan& t)
{
    static std::atomic<uint64_t> attr_mask{0};  // per device (omc_kernels.h)
    const size_t lds_bytes = sizeof(float) * (size_t)tile_lds_floats(H, L);
    if (lds_bytes > 48 * 1024) {
        hipError_t e = set_max_dynamic_lds(attr_mask, reinterpret_cast<const void*>(mlp_train_tile_kernel<H, L>), lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (!t.wt_current) hipLaunchKernelGGL(mlp_transpose_kernel, dim3(64), dim3(256), 0, st, t.params, t.wt, H, L);
    const Shuffle sh = plan_shuffle(t);
    int64_t step = t.first_step;
    const int64_t nsteps = plan_steps(t);
    for (int64_t k = 0; k < nsteps; ++k) {
        const StepSpan sp = plan_span(t, k);
        const int64_t o = sp.row0, nb = sp.local;
        ++step;
        MlpTileArgs a;
        a.data = t.data;
        a.params = t.params;
        a.wt = t.wt;
        a.partial = t.partial;
        a.row0 = o;
        a.nrows = nb;
        a.shuf = sh;
        a.ntiles = (int)((nb + 31) / 32);
        a.pstride = tile_pstride(H, L);
        a.two_over_b = (float)(2.0 / (double)sp.global);
        a.drop_pos = t.drop_pos ? t.drop_pos + o : nullptr;
        a.keep16 = t.dropout > 0.0 ? (uint32_t)llround((1.0 - t.dropout) * 65536.0) : 65536u;
        a.inv_keep = a.keep16 >= 65536u ? 1.0f : (float)(65536.0 / (double)a.keep16);
        a.step = (uint32_t)step;
        a.k0 = (uint32_t)t.seed;
        a.k1 = (uint32_t)(t.seed >> 32);
        const int waves = a.ntiles < tile_waves_max(H) ? a.ntiles : tile_waves_max(H);
        for (int t0 = 0; t0 < a.ntiles; t0 += waves) {
            a.tile0 = t0;
            a.accumulate = t0 > 0;
            hipLaunchKernelGGL((mlp_train_tile_kernel<H, L>), dim3(waves), dim3(64), lds_bytes, st, a);
        }
        MlpAdamArgs b;
        b.params = t.params;
        b.m = t.adam_m;
        b.v = t.adam_v;
        b.partial = t.partial;
        b.loss_acc = t.loss_acc;
        b.nparts = waves;
        b.nparams = mlp_params_of(H, L);
        b.stride = a.pstride;
        b.wt = t.wt; b.H = H; b.L = L;
        b.inv_b = (float)(1.0 / (double)sp.global);
        const double bc1 = 1.0 - pow(t.beta1, (double)step), bc2 = 1.0 - pow(t.beta2, (double)step);
        b.lr_t = (float)(t.lr / bc1);
        b.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        b.beta1 = (float)t.beta1;
        b.beta2 = (float)t.beta2;
        b.eps = (float)t.eps;
        b.wd = (float)t.weight_decay;
        if (finish_step(st, t, b, (mlp_params_of(H, L) + 16) / 16)) return hipErrorUnknown;
    }
    return hipGetLastError();
}

// minibatches of at most kMlpMaxGroups tiles: one workgroup per tile (mlp_train_quad_kernel: 32-row tiles;
// mlp_train_q16_kernel: 16-row tiles)
template <int H, int L, bool Q16 = false>
static hipError_t quad_steps(hipStream_t st, const MlpTrainPlan& t)
{
    constexpr int kTileRows = Q16 ? 16 : 32;
    if (!t.wt_current) hipLaunchKernelGGL(mlp_transpose_kernel, dim3(64), dim3(256), 0, st, t.params, t.wt, H, L);
    const Shuffle sh = plan_shuffle(t);
    int64_t step = t.first_step;
    const int64_t nsteps = plan_steps(t);
    for (int64_t k = 0; k < nsteps; ++k) {
        const StepSpan sp = plan_span(t, k);
        const int64_t o = sp.row0, nb = sp.local;
        ++step;
        MlpQuadArgs a;
        a.data = t.data;
        a.params = t.params;
        a.wt = t.wt;
        a.partial = t.partial;
        a.row0 = o;
        a.nrows = nb;
        a.shuf = sh;
        a.ntiles = (int)((nb + kTileRows - 1) / kTileRows);
        a.pstride = tile_pstride(H, L);
        a.two_over_b = (float)(2.0 / (double)sp.global);
        a.drop_pos = t.drop_pos ? t.drop_pos + o : nullptr;
        a.keep16 = t.dropout > 0.0 ? (uint32_t)llround((1.0 - t.dropout) * 65536.0) : 65536u;
        a.inv_keep = a.keep16 >= 65536u ? 1.0f : (float)(65536.0 / (double)a.keep16);
        a.step = (uint32_t)step;
        a.k0 = (uint32_t)t.seed;
        a.k1 = (uint32_t)(t.seed >> 32);
        if constexpr (Q16) {
            if (a.ntiles > 0) hipLaunchKernelGGL((mlp_train_q16_kernel<H, L>), dim3(a.ntiles), dim3(H * 2), 0, st, a);
        } else {
            if (a.ntiles > 0) hipLaunchKernelGGL((mlp_train_quad_kernel<H, L>), dim3(a.ntiles), dim3(H * 2), 0, st, a);
        }
        MlpAdamArgs b;
        b.params = t.params;
        b.m = t.adam_m;
        b.v = t.adam_v;
        b.partial = t.partial;
        b.loss_acc = t.loss_acc;
        b.nparts = a.ntiles;
        b.nparams = mlp_params_of(H, L);
        b.stride = a.pstride;
        b.wt = t.wt; b.H = H; b.L = L;
        b.inv_b = (float)(1.0 / (double)sp.global);
        const double bc1 = 1.0 - pow(t.beta1, (double)step), bc2 = 1.0 - pow(t.beta2, (double)step);
        b.lr_t = (float)(t.lr / bc1);
        b.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        b.beta1 = (float)t.beta1;
        b.beta2 = (float)t.beta2;
        b.eps = (float)t.eps;
        b.wd = (float)t.weight_decay;
        if (finish_step(st, t, b, (mlp_params_of(H, L) + 16) / 16)) return hipErrorUnknown;
    }
    return hipGetLastError();
}

// The minibatch size that picks the trainer kernel and sizes its partial buffer: the GLOBAL one, sharded or not.  A rank's
// own part of a minibatch is never larger (omc_mlp_train_epoch_sharded checks step_off against it), so the buffer covers
// it; and every rank of a job -- and the unsharded run of the same job -- picks the SAME kernel, hence the same dropout
// tags and unit map: a rank's masks are those of the unsharded run whatever its share of a minibatch is (a choice by the
// local share used to split ranks at the 4,096-row limit of the 16-row kernel: ADVICE r5).
int64_t mlp_plan_kernel_batch(const MlpTrainPlan& t) { return t.batch; }

hipError_t mlp_train_steps(hipStream_t st, const MlpTrainPlan& t)
{
    int choice = mlp_train_kernel_choice(t.hidden, t.layers, mlp_plan_kernel_batch(t));
    if (choice == 4 && !t.allow_q16) choice = 3;  // (4 is only ever returned where 3 applies)
    if (choice == 1) return t.layers == 2 ? train_steps<2>(st, t) : train_steps<3>(st, t);
    if (choice == 2) {
        if (t.hidden == 64) return t.layers == 2 ? tile_steps<64, 2>(st, t) : tile_steps<64, 3>(st, t);
        return t.layers == 2 ? tile_steps<128, 2>(st, t) : tile_steps<128, 3>(st, t);
    }
    if (choice == 3) {
        if (t.hidden == 32) return t.layers == 2 ? quad_steps<32, 2>(st, t) : quad_steps<32, 3>(st, t);
        if (t.hidden == 64) return t.layers == 2 ? quad_steps<64, 2>(st, t) : quad_steps<64, 3>(st, t);
        return t.layers == 2 ? quad_steps<128, 2>(st, t) : quad_steps<128, 3>(st, t);
    }
    if (choice == 4) {
        if (t.hidden == 64) return t.layers == 2 ? quad_steps<64, 2, true>(st, t) : quad_steps<64, 3, true>(st, t);
        return t.layers == 2 ? quad_steps<128, 2, true>(st, t) : quad_steps<128, 3, true>(st, t);
    }
    return hipErrorInvalidValue;
}

// ---- batched training: one launch pair per step for ALL problems of a batch (mlp_train_quad_batch_kernel)
size_t mlp_batch_table_bytes(int n) { return sizeof(MlpBatchProb) * (size_t)n; }

bool mlp_batch_supported(int hidden, int layers, int64_t batch)
{
    return mlp_train_kernel_choice32(hidden, layers, batch) == 3;  // the one-tile-per-workgroup trainer (32-row tiles)
}

void mlp_batch_table_image(const MlpBatchJob* jobs, int n, int hidden, int layers, double beta1, double beta2, double eps,
                           double weight_decay, double dropout, void* out)
{
    MlpBatchProb* tab = (MlpBatchProb*)out;
    for (int i = 0; i < n; ++i) {
        const MlpBatchJob& j = jobs[i];
        MlpBatchProb& p = tab[i];
        memset(&p, 0, sizeof p);
        p.data = j.data; p.params = j.params; p.m = j.adam_m; p.v = j.adam_v; p.partial = j.partial; p.wt = j.wt;
        p.loss_acc = j.loss_acc;
        p.nrows = j.nrows; p.batch = j.batch; p.first_step = j.first_step;
        p.lr = j.lr; p.beta1 = beta1; p.beta2 = beta2; p.eps = eps; p.wd = weight_decay;
        p.shuf = make_shuffle(j.nrows, j.shuffle_key);
        p.keep16 = dropout > 0.0 ? (uint32_t)llround((1.0 - dropout) * 65536.0) : 65536u;
        p.inv_keep = p.keep16 >= 65536u ? 1.0f : (float)(65536.0 / (double)p.keep16);
        p.k0 = (uint32_t)j.seed;
        p.k1 = (uint32_t)(j.seed >> 32);
        p.pstride = tile_pstride(hidden, layers);
        p.q16_rows = j.allow_q16 ? (int)mlp_q16_rows(hidden) : 0;
        p.nrows_dev = j.nrows_dev;
    }
}

template <int H, int L>
static hipError_t batch_epoch(hipStream_t st, const MlpBatchProb* tab, int n, int64_t max_steps, int max_tiles32,
                              int max_tiles16, const double* bc1, const double* bc2)
{
    hipLaunchKernelGGL(mlp_transpose_batch_kernel, dim3(16, n), dim3(256), 0, st, tab, H, L);
    const bool flat = n >= 8;
    const dim3 ga((unsigned)(flat ? (mlp_params_of(H, L) + 256) / 256 : (mlp_params_of(H, L) + 16) / 16), (unsigned)n);
    for (int64_t s = 0; s < max_steps; ++s) {
        // every problem runs the kernel its own omc_mlp_train_epoch call runs: one launch for the problems in 32-row
        // tiles, one for those in 16-row tiles (a curve's points normally all share the reference's minibatch of 256)
        if (max_tiles32 > 0)
            hipLaunchKernelGGL((mlp_train_quad_batch_kernel<H, L, false>), dim3((unsigned)max_tiles32, (unsigned)n), dim3(H * 2), 0,
                               st, tab, (int)s, 0);
        if constexpr (H >= 64) {
            if (max_tiles16 > 0)
                hipLaunchKernelGGL((mlp_train_quad_batch_kernel<H, L, true>), dim3((unsigned)max_tiles16, (unsigned)n), dim3(H * 2),
                                   0, st, tab, (int)s, 0);
        }
        if (flat) hipLaunchKernelGGL(mlp_adam_batch_kernel<true>, ga, dim3(256), 0, st, tab, (int)s, 0, H, L, bc1, bc2);
        else hipLaunchKernelGGL(mlp_adam_batch_kernel<false>, ga, dim3(256), 0, st, tab, (int)s, 0, H, L, bc1, bc2);
    }
    return hipGetLastError();
}

// ONE full-batch optimizer step (forward / backward + Adam) for every problem of the table: the per-step ContNet
// flow's "epoch".  `step_base` = optimizer steps the nets have taken so far (Adam's bias correction); the transposed
// connection copies must be current (the flow's init kernel writes them, Adam keeps them so).
template <int H, int L>
static hipError_t batch_one_step(hipStream_t st, const MlpBatchProb* tab, int n, int grid_tiles, int step_base,
                                 const double* bc1, const double* bc2, const int* prefix)
{
    const bool flat = n >= 8;
    const dim3 gq((unsigned)grid_tiles, (unsigned)n),
        ga((unsigned)(flat ? (mlp_params_of(H, L) + 256) / 256 : (mlp_params_of(H, L) + 16) / 16), (unsigned)n);
    if (prefix) hipLaunchKernelGGL((mlp_train_quad_list_kernel<H, L>), dim3((unsigned)grid_tiles), dim3(H * 2), 0, st, tab, prefix, n, step_base);
    else hipLaunchKernelGGL((mlp_train_quad_batch_kernel<H, L, false>), gq, dim3(H * 2), 0, st, tab, 0, step_base);
    if (flat) hipLaunchKernelGGL(mlp_adam_batch_kernel<true>, ga, dim3(256), 0, st, tab, 0, step_base, H, L, bc1, bc2);
    else hipLaunchKernelGGL(mlp_adam_batch_kernel<false>, ga, dim3(256), 0, st, tab, 0, step_base, H, L, bc1, bc2);
    return hipGetLastError();
}

hipError_t mlp_train_step_batch(hipStream_t st, const void* table_dev, int n, int hidden, int grid_tiles, int step_base,
                                const double* bc1_dev, const double* bc2_dev, const int* tile_prefix_dev)
{
    const MlpBatchProb* tab = (const MlpBatchProb*)table_dev;
    if (hidden == 32) return batch_one_step<32, 2>(st, tab, n, grid_tiles, step_base, bc1_dev, bc2_dev, tile_prefix_dev);
    if (hidden == 64) return batch_one_step<64, 2>(st, tab, n, grid_tiles, step_base, bc1_dev, bc2_dev, tile_prefix_dev);
    if (hidden == 128) return batch_one_step<128, 2>(st, tab, n, grid_tiles, step_base, bc1_dev, bc2_dev, tile_prefix_dev);
    return hipErrorInvalidValue;
}

hipError_t mlp_tile_prefix(hipStream_t st, const void* table_dev, int n, int* prefix_dev)
{
    hipLaunchKernelGGL(mlp_tile_prefix_kernel, dim3(1), dim3(1024), 0, st, (const MlpBatchProb*)table_dev, n, prefix_dev);
    return hipGetLastError();
}

hipError_t mlp_train_epoch_batch(hipStream_t st, const void* table_dev, int n, int hidden, int layers, int64_t max_steps,
                                 int max_tiles32, int max_tiles16, const double* bc1_dev, const double* bc2_dev)
{
    const MlpBatchProb* tab = (const MlpBatchProb*)table_dev;
    const int t32 = max_tiles32, t16 = max_tiles16;
    if (hidden == 32) return layers == 2 ? batch_epoch<32, 2>(st, tab, n, max_steps, t32, 0, bc1_dev, bc2_dev)
                                         : batch_epoch<32, 3>(st, tab, n, max_steps, t32, 0, bc1_dev, bc2_dev);
    if (hidden == 64) return layers == 2 ? batch_epoch<64, 2>(st, tab, n, max_steps, t32, t16, bc1_dev, bc2_dev)
                                         : batch_epoch<64, 3>(st, tab, n, max_steps, t32, t16, bc1_dev, bc2_dev);
    if (hidden == 128) return layers == 2 ? batch_epoch<128, 2>(st, tab, n, max_steps, t32, t16, bc1_dev, bc2_dev)
                                          : batch_epoch<128, 3>(st, tab, n, max_steps, t32, t16, bc1_dev, bc2_dev);
    return hipErrorInvalidValue;
}

int mlp_apply_param_count(int hidden, int layers)
{
    if ((hidden != 32 && hidden != 64 && hidden != 128) || (layers != 2 && layers != 3)) return -1;
    return mlp_params_of(hidden, layers);
}

template <int H, int L>
static hipError_t launch_apply(hipStream_t st, const MlpApplyArgs& a)
{
    static std::atomic<uint64_t> attr_mask{0};  // per device (omc_kernels.h)
    const size_t lds_bytes = sizeof(float) * (size_t)apply_lds_floats(H, L);
    if (true) {
        hipError_t e = set_max_dynamic_lds(attr_mask, reinterpret_cast<const void*>(mlp_apply_kernel<H, L>), lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((mlp_apply_kernel<H, L>), dim3((unsigned)((a.ntiles + 3) / 4)), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}

hipError_t mlp_apply_pass2(hipStream_t st, const LsmProblem& p, int hidden, int layers, const float* params,
                           const double* feat_mean, const double* feat_std, double y_mean, double y_std,
                           double dropout, uint64_t seed, float* sx, int32_t* tex, int64_t col_base0, int64_t col_base1)
{
    MlpApplyArgs a;
    a.key_half = p.M / 2; a.key_base0 = col_base0; a.key_base1 = col_base1;
    a.S = p.S; a.ld = p.ld; a.M = p.M; a.N = p.N; a.is_put = p.is_put;
    a.K = p.K; a.T = p.T; a.dt = p.T / (double)p.N;
    a.params = params;
    for (int i = 0; i < 7; ++i) {
        a.fm[i] = feat_mean[i];
        a.rs[i] = 1.0 / feat_std[i];
    }
    a.ym = y_mean; a.ysd = y_std;
    a.sx = sx; a.tex = tex;
    a.keep16 = dropout > 0.0 ? (uint32_t)llround((1.0 - dropout) * 65536.0) : 65536u;
    a.inv_keep = a.keep16 >= 65536u ? 1.0f : (float)(65536.0 / (double)a.keep16);
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
    a.ntiles = (int)((p.M + 31) / 32);
    if (hidden == 32 && layers == 2) return launch_apply<32, 2>(st, a);
    if (hidden == 32 && layers == 3) return launch_apply<32, 3>(st, a);
    if (hidden == 64 && layers == 2) return launch_apply<64, 2>(st, a);
    if (hidden == 64 && layers == 3) return launch_apply<64, 3>(st, a);
    if (hidden == 128 && layers == 2) return launch_apply<128, 2>(st, a);
    if (hidden == 128 && layers == 3) return launch_apply<128, 3>(st, a);
    return hipErrorInvalidValue;
}

int localvol_param_count(int hidden, int layers)
{
    if (hidden != 64 || layers < 1 || layers > 8) return -1;
    return kH * 4 + layers * (kH * kH + 3 * kH) + kH + 1;
}

hipError_t localvol_paths(hipStream_t st, float* S, int64_t ld, int64_t M, int N, int layers, const float* params,
                          const float* Z, double S0, double r, double T, double K, double m_scale,
                          double tau_scale, double eps_out)
{
    LocalVolArgs a;
    a.S = S; a.ld = ld; a.M = M; a.P = M / 2; a.N = N; a.L = layers;
    a.params = params; a.Z = Z;
    const double dt = T / (double)N;
    a.s0 = (float)S0; a.r = (float)r; a.dt = (float)dt; a.sqdt = (float)sqrt(dt); a.eps_out = (float)eps_out;
    a.K = K; a.T = T; a.dtd = dt; a.inv_m_scale = 1.0 / m_scale; a.inv_tau_scale = 1.0 / tau_scale;
    a.ntiles = (int)((M + 31) / 32);
    const size_t lds_bytes = sizeof(float) * (size_t)(kH * 4 + layers * (kH * kLdW2 + 3 * kH) + kH + 4);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(localvol_paths_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(localvol_paths_kernel, dim3((unsigned)((a.ntiles + 3) / 4)), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}

}  // namespace omc
