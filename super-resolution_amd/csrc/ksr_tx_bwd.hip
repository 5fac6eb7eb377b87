// Backward of the batch-axis transformer layer (see ksr_tx.hip for the forward and the lane layout).
//
//   tx_bwd_token_kernel : everything that is per token - [final LayerNorm] <- LayerNorm2 <- MLP <-
//                         LayerNorm1 <- out-proj; writes the gradient tensors the parameter-gradient
//                         reductions (ksr_wgrad / ksr_colreduce) consume, and d(ctx).
//   tx_bwd_attn_kernel  : softmax-attention backward over the S images of a pixel: d(q|k|v).
// The QKV projection's data gradient (+ residual) is ksr_in_proj_bwd (ksr_bwd.hip).
#include "ksr_gemm.h"
#include "ksr_wg_fuse.h"

// gradient sums of up to three LayerNorms from per-workgroup rows (defined below)
int ksr_ln_reduce(const float* slab, int G, float* out0, float* out1, float* out2, int row_stride, hipStream_t stream);

namespace {

struct TxBwdTokArgs {
    const float* dy;                          // [N,64] gradient of the layer output
    const float *xhatf, *rstdf, *nf_w;        // final LayerNorm (optional: nf_w == null -> absent)
    const float *xhat2, *rstd2, *n2_w;
    const float* hmid;                        // [N,32] (after ReLU and dropout)
    const float *xhat1, *rstd1, *n1_w;
    const float *l2_w, *l1_w, *out_w;         // [64,32], [32,64], [64,64]
    float *ds2, *dm, *dpre, *ds1, *du1, *da, *dctx;   // outputs (ds2 only with the final LayerNorm)
    float* ln_slab;                           // optional [3][gridDim.x][2][64]: per-workgroup LayerNorm gain / shift sums
    long N;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_dev;   // optional device word mixed into `seed` at kernel start (ksr_set_dropout_seed_source)
};

struct __attribute__((aligned(16))) TxBwdTokSmem {
    float f_l2t[1 * 32 * 64];      // 64 -> 32
    float f_l1t[2 * 16 * 64];      // 32 -> 64
    float f_outt[2 * 32 * 64];     // 64 -> 64
    float ln[3][64];
    float xpose[4 * KSR_XPOSE_FLOATS];   // per-wave transposers of the coalesced loads / stores
};

__device__ __forceinline__ void ln_bwd(const float (&dy)[32], const float* __restrict__ gain, int h,
                                       const float (&xhat)[32], float rstd, float (&dx)[32]) {
    float dxh[32], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        dxh[i] = dy[i] * gain[32 * h + i];
        s1 += dxh[i];
        s2 += dxh[i] * xhat[i];
    }
    const float m1 = ksr_pair_sum(s1) * (1.f / 64.f), m2 = ksr_pair_sum(s2) * (1.f / 64.f);
#pragma unroll
    for (int i = 0; i < 32; ++i) dx[i] = rstd * (dxh[i] - m1 - xhat[i] * m2);
}

// the same in place: g (gradient of the LayerNorm's output) becomes the gradient of its input
__device__ __forceinline__ void ln_bwd_inplace(float (&g)[32], const float* __restrict__ gain, int h, const float (&xhat)[32], float rstd) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        g[i] *= gain[32 * h + i];
        s1 += g[i];
        s2 += g[i] * xhat[i];
    }
    const float m1 = ksr_pair_sum(s1) * (1.f / 64.f), m2 = ksr_pair_sum(s2) * (1.f / 64.f);
#pragma unroll
    for (int i = 0; i < 32; ++i) g[i] = rstd * (g[i] - m1 - xhat[i] * m2);
}

template <int PREC>
__global__ void __launch_bounds__(256, 2) tx_bwd_token_kernel(TxBwdTokArgs a) {
    KSR_MIX_SEED(a);
    using ST = typename KsrStore<PREC>::T;
    __shared__ TxBwdTokSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, 32, 1, true>(sm.f_l2t, nullptr, a.l2_w, 32, nullptr, MapSplit{16}, MapHL{32}, tid, 256);
    ksr_stage_weights<PREC, 16, 2, true>(sm.f_l1t, nullptr, a.l1_w, 64, nullptr, MapHL{32}, MapSplit{16}, tid, 256);
    ksr_stage_weights<PREC, 32, 2, true>(sm.f_outt, nullptr, a.out_w, 64, nullptr, MapHL{32}, MapHL{32}, tid, 256);
    if (tid < 64) {
        sm.ln[0][tid] = a.nf_w ? a.nf_w[tid] : 1.f;
        sm.ln[1][tid] = a.n2_w[tid];
        sm.ln[2][tid] = a.n1_w[tid];
    }
    __syncthreads();
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const bool drop = a.p_drop > 0.f;
    const long ntiles = (a.N + 127) / 128;
    const bool ln_sums = a.ln_slab != nullptr;
    float cs[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};     // lane = channel: [final, norm2, norm1][sum g*xhat, sum g]
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0 = tile * 128 + wv * 32 + tok;
        const bool live = n0 < a.N;
        const long n = live ? n0 : 0;
        const long n0w = tile * 128 + wv * 32;             // the wave's 32 consecutive rows
        const int nvalid = (a.N - n0w) < 32 ? (int)(a.N - n0w > 0 ? a.N - n0w : 0) : 32;
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;

        // bf16 storage: the inputs of the tile are requested ahead of their use (loads complete in issue order, so
        // each stage waits only for its own), which keeps several tiles of loads in flight per wave; fp32 tiles are
        // twice as many registers and are fetched where they are used
        constexpr bool EARLY = sizeof(ST) == 2;
        KsrTile<ST> t_dy, t_xf, t_x2, t_x1;
        KsrTile16<ST> t_hm;
        ksr_co_fetch<ST>(a.dy, n0w, nvalid, 64, 0, lane, t_dy);
        if (EARLY) {
            if (a.nf_w) ksr_co_fetch<ST>(a.xhatf, n0w, nvalid, 64, 0, lane, t_xf);
            ksr_co_fetch<ST>(a.xhat2, n0w, nvalid, 64, 0, lane, t_x2);
            ksr_co16_fetch<ST>(a.hmid, n0w, nvalid, lane, t_hm);
        }
        const float rs_f = a.nf_w ? a.rstdf[n] : 0.f, rs_2 = a.rstd2[n], rs_1 = a.rstd1[n];
        float g[32], xh[32];
        ksr_co_finish<ST>(xb, lane, t_dy, g);
        if (a.nf_w) {
            float t[32];
            if (!EARLY) ksr_co_fetch<ST>(a.xhatf, n0w, nvalid, 64, 0, lane, t_xf);
            ksr_co_finish<ST>(xb, lane, t_xf, xh);
            if (ln_sums) ksr_colsum32(xb, lane, g, xh, cs[0][0], cs[0][1]);
            ln_bwd(g, sm.ln[0], h, xh, rs_f, t);
#pragma unroll
            for (int i = 0; i < 32; ++i) g[i] = t[i];
            if (a.ds2) ksr_store32_co<ST>(xb, a.ds2, n0w, nvalid, 64, 0, lane, g);
        }
        // LayerNorm 2
        float du2[32];
        if (!EARLY) ksr_co_fetch<ST>(a.xhat2, n0w, nvalid, 64, 0, lane, t_x2);
        ksr_co_finish<ST>(xb, lane, t_x2, xh);
        if (ln_sums) ksr_colsum32(xb, lane, g, xh, cs[1][0], cs[1][1]);
        ln_bwd(g, sm.ln[1], h, xh, rs_2, du2);
        // MLP
        float dm[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            dm[i] = drop ? (ksr_keep(a.seed, 3u, (unsigned long long)n * 64 + 32 * h, i, keep) ? du2[i] * inv_keep : 0.f)
                         : du2[i];
        ksr_store32_co<ST>(xb, a.dm, n0w, nvalid, 64, 0, lane, dm);
        if (EARLY) ksr_co_fetch<ST>(a.xhat1, n0w, nvalid, 64, 0, lane, t_x1);
        float dpre[16];
        {
            f32x16 acc[1];
            ksr_wave_gemm<PREC, 32, 1>(sm.f_l2t, nullptr, dm, acc, lane);
            float hm[16];
            if (!EARLY) ksr_co16_fetch<ST>(a.hmid, n0w, nvalid, lane, t_hm);
            ksr_co16_finish<ST>(xb, lane, t_hm, hm);
#pragma unroll
            for (int i = 0; i < 16; ++i) dpre[i] = hm[i] > 0.f ? (drop ? acc[0][i] * inv_keep : acc[0][i]) : 0.f;
            ksr_store16_co<ST>(xb, a.dpre, n0w, nvalid, lane, dpre);
        }
        float ds1[32];
        {
            f32x16 acc[2];
            ksr_wave_gemm<PREC, 16, 2>(sm.f_l1t, nullptr, dpre, acc, lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) { ds1[i] = du2[i] + acc[0][i]; ds1[16 + i] = du2[16 + i] + acc[1][i]; }
            if (a.ds1) ksr_store32_co<ST>(xb, a.ds1, n0w, nvalid, 64, 0, lane, ds1);
        }
        // LayerNorm 1
        float du1[32];
        if (!EARLY) ksr_co_fetch<ST>(a.xhat1, n0w, nvalid, 64, 0, lane, t_x1);
        ksr_co_finish<ST>(xb, lane, t_x1, xh);
        if (ln_sums) ksr_colsum32(xb, lane, ds1, xh, cs[2][0], cs[2][1]);
        ln_bwd(ds1, sm.ln[2], h, xh, rs_1, du1);
        ksr_store32_co<ST>(xb, a.du1, n0w, nvalid, 64, 0, lane, du1);
        if (a.dctx == nullptr) continue;      // the out-projection's backward runs in a kernel of its own (ksr_batch_tx_layer_bwd_out_wg)
        float da[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            da[i] = drop ? (ksr_keep(a.seed, 1u, (unsigned long long)n * 64 + 32 * h, i, keep) ? du1[i] * inv_keep : 0.f)
                         : du1[i];
        if (drop) ksr_store32_co<ST>(xb, a.da, n0w, nvalid, 64, 0, lane, da);
        {
            f32x16 acc[2];
            ksr_wave_gemm<PREC, 32, 2>(sm.f_outt, nullptr, da, acc, lane);
            float dc[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) { dc[i] = acc[0][i]; dc[16 + i] = acc[1][i]; }
            ksr_store32_co<ST>(xb, a.dctx, n0w, nvalid, 64, 0, lane, dc);
        }
    }
    if (ln_sums) {
        // the four waves' sums meet in LDS (each wave's own transposer first, so no wave is still reading it)
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;
        ksr_wave_fence();
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int q = 0; q < 2; ++q) xb[(2 * l + q) * 64 + lane] = cs[l][q];
        __syncthreads();
        for (int i = tid; i < 6 * 64; i += 256) {
            const float v = (sm.xpose[i] + sm.xpose[KSR_XPOSE_FLOATS + i]) + (sm.xpose[2 * KSR_XPOSE_FLOATS + i] + sm.xpose[3 * KSR_XPOSE_FLOATS + i]);
            const int l = i / 128, r = i - 128 * l;
            a.ln_slab[((long)l * gridDim.x + blockIdx.x) * 128 + r] = v;
        }
    }
}

// gain / shift gradients of up to three LayerNorms from the per-workgroup sums: out[l][q * row_stride + c]
struct LnRedArgs { const float* slab; int G; float* out[3]; int row_stride; };
__global__ void __launch_bounds__(1024) ln_reduce_kernel(LnRedArgs a) {
    // grid (3, 4): 32 of the 128 sums per workgroup x 32 slab groups, four independent partial sums per thread
    __shared__ float part[32][33];
    const int l = blockIdx.x;
    if (a.out[l] == nullptr) return;
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.y * 32 + o;
    const float* slab = a.slab + (long)l * a.G * 128 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = grp;
    for (; g + 96 < a.G; g += 128) {
        s0 += slab[(long)g * 128]; s1 += slab[(long)(g + 32) * 128];
        s2 += slab[(long)(g + 64) * 128]; s3 += slab[(long)(g + 96) * 128];
    }
    for (; g < a.G; g += 32) s0 += slab[(long)g * 128];
    part[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0) return;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += part[r][o];
    a.out[l][(e >> 6) * a.row_stride + (e & 63)] = s;
}

// ---------------------------------------------------------------------------------------------
// bf16 storage: the out-projection's backward as a kernel of its own that also accumulates out_proj's weight gradient
// (the scheme of ksr_wg_fuse.h: both operands of dW = da^T ctx are in its registers): the token kernel then stops at du1,
// and da (du1 under the attention branch's dropout mask), which only fed that product and this matrix product, is never
// written.  Tried and dropped: linear1's and linear2's weight gradients inside the token kernel.  64 more accumulator
// registers beside its 246 spill 200-330 registers in every arrangement of the prefetches; with the output blocks split over
// the four waves (operand tiles of all 128 tokens in LDS, 16 accumulators per wave, two barriers per tile) and only
// linear2's product it still spills 99 (0.53 ms per launch against 0.23 + 0.05 for the separate kernels), and built for one
// wave per SIMD (no spills) it takes 0.33 ms.
struct TxBwdOutArgs {
    const void* du1; const void* ctx;     // [N,64]
    const float* out_w;                   // [64][64]
    void* dctx;
    float* wg_slab;                       // [gridDim.x][KSR_WGF_ROW]
    long N;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_dev;
};

struct __attribute__((aligned(16))) TxBwdOutSmem {
    float f_outt[2 * 32 * 64 / 2];
    float xpose[4 * KSR_XPOSE_FLOATS];
};

__global__ void __launch_bounds__(256, 2) tx_bwd_out_wg_kernel(TxBwdOutArgs a) {
    KSR_MIX_SEED(a);
    using ST = __bf16;
    constexpr int PREC = KSR_BF16S;
    __shared__ TxBwdOutSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, 32, 2, true>(sm.f_outt, nullptr, a.out_w, 64, nullptr, MapHL{32}, MapHL{32}, tid, 256);
    __syncthreads();
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const bool drop = a.p_drop > 0.f;
    const long ntiles = (a.N + 127) / 128;
    KsrWgFused wgf;
    ksr_wgf_init(wgf);
    KsrTile<ST> t_du, t_cx;
    auto fetch = [&](long t) {
        const long r0 = t * 128 + wv * 32;
        const int nv = (a.N - r0) < 32 ? (int)(a.N - r0 > 0 ? a.N - r0 : 0) : 32;
        ksr_co_fetch<ST, true>(a.du1, r0, nv, 64, 0, lane, t_du);
        ksr_co_fetch<ST, true>(a.ctx, r0, nv, 64, 0, lane, t_cx);
    };
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0 = tile * 128 + wv * 32 + tok;
        const long n = n0 < a.N ? n0 : 0;
        const long n0w = tile * 128 + wv * 32;
        const int nvalid = (a.N - n0w) < 32 ? (int)(a.N - n0w > 0 ? a.N - n0w : 0) : 32;
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;
        float da[32], cx[32];
        ksr_co_finish<ST>(xb, lane, t_du, da);
        ksr_co_finish<ST>(xb, lane, t_cx, cx);
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);     // in flight during this tile
        if (drop) {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                da[i] = ksr_keep(a.seed, 1u, (unsigned long long)n * 64 + 32 * h, i, keep) ? da[i] * inv_keep : 0.f;
        }
        {
            f32x16 acc[2];
            ksr_wave_gemm<PREC, 32, 2>(sm.f_outt, nullptr, da, acc, lane);
            float dc[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) { dc[i] = acc[0][i]; dc[16 + i] = acc[1][i]; }
            ksr_store32_co<ST, false>(xb, a.dctx, n0w, nvalid, 64, 0, lane, dc);
        }
        ksr_wgf_accumulate(xb, lane, da, cx, wgf);     // rows past the end of the tensor hold zeros in da
    }
    ksr_wgf_finish(sm.xpose, a.wg_slab + (long)blockIdx.x * KSR_WGF_ROW, tid, wgf);
}

// ---------------------------------------------------------------------------------------------
// bf16 storage: the token backward in two kernels that each accumulate one of the MLP's weight gradients, so that the tensors
// which only fed those products are read where they are made:
//   tx_bwd_mlp_wg_kernel : LayerNorm 2 <- linear2 <- ReLU / dropout: writes du2 (the residual branch's
//                          gradient) and dpre, accumulates dW2 = dm^T hmid, db2 (dm = du2 under linear2's dropout mask is
//                          never written);
//   tx_bwd_ln1_wg_kernel : ds1 = du2 + dpre W1, LayerNorm 1 backward: writes du1, accumulates dW1 = dpre^T xhat1, db1.
// Against tx_bwd_token_kernel + two ksr_wgrad launches: 200 bytes per token and layer less (dm and dpre are not re-read,
// hmid and xhat1 are read once), and each kernel holds 32 accumulator registers where the one-kernel form would have
// needed 64 beside its 215 (it spilled in every arrangement).
struct TxBwdMlpArgs {
    const void* dy;
    const void* xhat2; const float *rstd2, *n2_w;
    const void* hmid;                                   // [N,32]
    const float* l2_w;                                  // [64,32]
    void *du2, *dpre;                                   // [N,64], [N,32]
    float* ln_slab;                                     // [gridDim.x][2][64]: norm2
    float* wg_slab;                                     // [gridDim.x][64*32 + 64]
    long N; float p_drop; unsigned long long seed; const unsigned long long* seed_dev;
};
struct __attribute__((aligned(16))) TxBwdMlpSmem {
    float f_l2t[1 * 32 * 64 / 2];
    float ln[64];
    float xpose[4 * KSR_XPOSE_FLOATS];
};

__global__ void __launch_bounds__(256, 2) tx_bwd_mlp_wg_kernel(TxBwdMlpArgs a) {
    // (A variant with the encoder's final LayerNorm backward in front spilled 61 registers in every arrangement - 0.36 ms per
    // launch against 0.125: that backward runs inside the feed + KAN backward instead, ksr_feed_kan_bwd_lnf.)
    KSR_MIX_SEED(a);
    using ST = __bf16;
    constexpr int PREC = KSR_BF16S;
    __shared__ TxBwdMlpSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, 32, 1, true>(sm.f_l2t, nullptr, a.l2_w, 32, nullptr, MapSplit{16}, MapHL{32}, tid, 256);
    if (tid < 64) sm.ln[tid] = a.n2_w[tid];
    __syncthreads();
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const bool drop = a.p_drop > 0.f;
    const long ntiles = (a.N + 127) / 128;
    float cs0 = 0.f, cs1 = 0.f;
    KsrWgFusedT<2, 1> wgf;
    ksr_wgft_init(wgf);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0 = tile * 128 + wv * 32 + tok;
        const long n = n0 < a.N ? n0 : 0;
        const long n0w = tile * 128 + wv * 32;
        const int nvalid = (a.N - n0w) < 32 ? (int)(a.N - n0w > 0 ? a.N - n0w : 0) : 32;
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;
        KsrTile<ST> t_dy, t_x2;
        KsrTile16<ST> t_hm;
        ksr_co_fetch<ST>(a.dy, n0w, nvalid, 64, 0, lane, t_dy);
        ksr_co_fetch<ST>(a.xhat2, n0w, nvalid, 64, 0, lane, t_x2);
        const float rs_2 = a.rstd2[n];
        float g[32], xh[32];
        ksr_co_finish<ST>(xb, lane, t_dy, g);
        ksr_co16_fetch<ST>(a.hmid, n0w, nvalid, lane, t_hm);
        ksr_co_finish<ST>(xb, lane, t_x2, xh);
        ksr_colsum32(xb, lane, g, xh, cs0, cs1);
        ln_bwd_inplace(g, sm.ln, h, xh, rs_2);                     // g = du2
        ksr_store32_co<ST>(xb, a.du2, n0w, nvalid, 64, 0, lane, g);
        float dm[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            dm[i] = drop ? (ksr_keep(a.seed, 3u, (unsigned long long)n * 64 + 32 * h, i, keep) ? g[i] * inv_keep : 0.f) : g[i];
        float hm[16], dpre[16];
        ksr_co16_finish<ST>(xb, lane, t_hm, hm);
        {
            f32x16 acc[1];
            ksr_wave_gemm<PREC, 32, 1>(sm.f_l2t, nullptr, dm, acc, lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) dpre[i] = hm[i] > 0.f ? (drop ? acc[0][i] * inv_keep : acc[0][i]) : 0.f;
        }
        ksr_store16_co<ST>(xb, a.dpre, n0w, nvalid, lane, dpre);
        ksr_wgft_accumulate<2, 1>(xb, lane, dm, hm, wgf);      // rows past the end of the tensor hold zeros in dm
    }
    {
        // LayerNorm gain / shift sums: the four waves' sums meet in LDS (each wave's own transposer first)
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;
        ksr_wave_fence();
        xb[lane] = cs0; xb[64 + lane] = cs1;
        __syncthreads();
        for (int i = tid; i < 128; i += 256)
            a.ln_slab[(long)blockIdx.x * 128 + i] =
                (sm.xpose[i] + sm.xpose[KSR_XPOSE_FLOATS + i]) + (sm.xpose[2 * KSR_XPOSE_FLOATS + i] + sm.xpose[3 * KSR_XPOSE_FLOATS + i]);
    }
    ksr_wgft_finish<2, 1>(sm.xpose, a.wg_slab + (long)blockIdx.x * (64 * 32 + 64), tid, wgf);
}

struct TxBwdLn1Args {
    const void *du2, *dpre, *xhat1; const float *rstd1, *n1_w;
    const float* l1_w;                                  // [32,64]
    void* du1;
    float* ln_slab;                                     // [gridDim.x][2][64]: norm1
    float* wg_slab;                                     // [gridDim.x][32*64 + 32]
    long N;
};
struct __attribute__((aligned(16))) TxBwdLn1Smem {
    float f_l1t[2 * 16 * 64 / 2];
    float ln[64];
    float xpose[4 * KSR_XPOSE_FLOATS];
};

__global__ void __launch_bounds__(256, 2) tx_bwd_ln1_wg_kernel(TxBwdLn1Args a) {
    using ST = __bf16;
    constexpr int PREC = KSR_BF16S;
    __shared__ TxBwdLn1Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, 16, 2, true>(sm.f_l1t, nullptr, a.l1_w, 64, nullptr, MapHL{32}, MapSplit{16}, tid, 256);
    if (tid < 64) sm.ln[tid] = a.n1_w[tid];
    __syncthreads();
    const long ntiles = (a.N + 127) / 128;
    float cs0 = 0.f, cs1 = 0.f;
    KsrWgFusedT<1, 2> wgf;
    ksr_wgft_init(wgf);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0 = tile * 128 + wv * 32 + tok;
        const long n = n0 < a.N ? n0 : 0;
        const long n0w = tile * 128 + wv * 32;
        const int nvalid = (a.N - n0w) < 32 ? (int)(a.N - n0w > 0 ? a.N - n0w : 0) : 32;
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;
        KsrTile<ST> t_du, t_x1;
        KsrTile16<ST> t_dp;
        ksr_co_fetch<ST>(a.du2, n0w, nvalid, 64, 0, lane, t_du);
        ksr_co16_fetch<ST>(a.dpre, n0w, nvalid, lane, t_dp);
        ksr_co_fetch<ST>(a.xhat1, n0w, nvalid, 64, 0, lane, t_x1);
        const float rs_1 = a.rstd1[n];
        float du2[32], dpre[16], xh[32];
        ksr_co_finish<ST>(xb, lane, t_du, du2);
        ksr_co16_finish<ST>(xb, lane, t_dp, dpre);
        float ds1[32];
        {
            f32x16 acc[2];
            ksr_wave_gemm<PREC, 16, 2>(sm.f_l1t, nullptr, dpre, acc, lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) { ds1[i] = du2[i] + acc[0][i]; ds1[16 + i] = du2[16 + i] + acc[1][i]; }
        }
        ksr_co_finish<ST>(xb, lane, t_x1, xh);
        ksr_wgft_accumulate<1, 2>(xb, lane, dpre, xh, wgf);     // rows past the end of the tensor hold zeros in dpre
        ksr_colsum32(xb, lane, ds1, xh, cs0, cs1);
        ln_bwd_inplace(ds1, sm.ln, h, xh, rs_1);                   // ds1 = du1
        ksr_store32_co<ST>(xb, a.du1, n0w, nvalid, 64, 0, lane, ds1);
    }
    {
        float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;
        ksr_wave_fence();
        xb[lane] = cs0; xb[64 + lane] = cs1;
        __syncthreads();
        for (int i = tid; i < 2 * 64; i += 256)
            a.ln_slab[(long)blockIdx.x * 128 + i] = (sm.xpose[i] + sm.xpose[KSR_XPOSE_FLOATS + i]) + (sm.xpose[2 * KSR_XPOSE_FLOATS + i] + sm.xpose[3 * KSR_XPOSE_FLOATS + i]);
    }
    ksr_wgft_finish<1, 2>(sm.xpose, a.wg_slab + (long)blockIdx.x * (32 * 64 + 32), tid, wgf);
}

// ---------------------------------------------------------------------------------------------
struct TxBwdAttnArgs {
    const float* qkv;    // [N,192] saved by the forward (q not scaled)
    const float* ctx;    // [N,64]
    const float* dctx;   // [N,64]
    float* dqkv;         // [N,192]
    int S, HW;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_dev;   // optional device word mixed into `seed` at kernel start (ksr_set_dropout_seed_source)
};

constexpr int XLD = 140;     // floats per slot in the exchange buffer (2 x 68 + 4 pad)

template <class ST>   // storage type of qkv / ctx / dctx / dqkv
__global__ void __launch_bounds__(256, 2) tx_bwd_attn_kernel(TxBwdAttnArgs a) {
    KSR_MIX_SEED(a);
    __shared__ __attribute__((aligned(16))) float xb[128 * XLD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    const int S = a.S, HW = a.HW;
    const int PPW = 128 / S;
    const int ntiles = (HW + PPW - 1) / PPW;
    const int slot = wv * 32 + tok;
    const int b = slot / PPW, pl = slot - b * PPW;
    const float scale = 0.17677669529663687f;
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const bool drop = a.p_drop > 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p = tile * PPW + pl;
        const bool live = (b < S) && (p < HW);
        const long n = live ? ((long)b * HW + p) : 0;
        float* mine = &xb[slot * XLD + 68 * h];

        // ---- phase A: this lane is a QUERY.  exchange buffer holds k | v per (slot, head) ----
        float q[32], dO[32];
        float m = -INFINITY, l = 0.f, D = 0.f;
        {
            float kv[32];
            ksr_loadn<ST, 32>(a.qkv, n * 192 + 64 + 32 * h, kv);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(mine + 4 * c) = f32x4{kv[4 * c], kv[4 * c + 1], kv[4 * c + 2], kv[4 * c + 3]};
            ksr_loadn<ST, 32>(a.qkv, n * 192 + 128 + 32 * h, kv);
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(mine + 32 + 4 * c) = f32x4{kv[4 * c], kv[4 * c + 1], kv[4 * c + 2], kv[4 * c + 3]};
            ksr_loadn<ST, 32>(a.qkv, n * 192 + 32 * h, q);
            ksr_loadn<ST, 32>(a.dctx, n * 64 + 32 * h, dO);
            float o[32];
            ksr_loadn<ST, 32>(a.ctx, n * 64 + 32 * h, o);
#pragma unroll
            for (int i = 0; i < 32; ++i) { q[i] *= scale; D += dO[i] * o[i]; }
        }
        __syncthreads();
        for (int j = 0; j < S; ++j) {
            const float* kp = &xb[(j * PPW + pl) * XLD + 68 * h];
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 4 * c);
                sc += q[4 * c] * kk[0] + q[4 * c + 1] * kk[1] + q[4 * c + 2] * kk[2] + q[4 * c + 3] * kk[3];
            }
            const float mn = fmaxf(m, sc);
            l = l * expf(m - mn) + expf(sc - mn);
            m = mn;
        }
        const float inv_l = 1.f / l;
        float dq[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) dq[i] = 0.f;
        for (int j = 0; j < S; ++j) {
            const float* kp = &xb[(j * PPW + pl) * XLD + 68 * h];
            float sc = 0.f, dp = 0.f;
            f32x4 kk[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                kk[c] = *reinterpret_cast<const f32x4*>(kp + 4 * c);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(kp + 32 + 4 * c);
                sc += q[4 * c] * kk[c][0] + q[4 * c + 1] * kk[c][1] + q[4 * c + 2] * kk[c][2] + q[4 * c + 3] * kk[c][3];
                dp += dO[4 * c] * vv[0] + dO[4 * c + 1] * vv[1] + dO[4 * c + 2] * vv[2] + dO[4 * c + 3] * vv[3];
            }
            const float pj = expf(sc - m) * inv_l;
            if (drop) {
                dp = ksr_attn_keep(ksr_attn_drop_base(a.seed, (unsigned long long)n, h), j, keep) ? dp * inv_keep : 0.f;
            }
            const float dS = pj * (dp - D) * scale;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                dq[4 * c + 0] += dS * kk[c][0]; dq[4 * c + 1] += dS * kk[c][1];
                dq[4 * c + 2] += dS * kk[c][2]; dq[4 * c + 3] += dS * kk[c][3];
            }
        }
        if (live) ksr_storen<ST, 32>(a.dqkv, n * 192 + 32 * h, dq);

        // ---- phase B: this lane is a KEY.  exchange buffer holds q*scale | dO | m, 1/l, D ----
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            *reinterpret_cast<f32x4*>(mine + 4 * c) = f32x4{q[4 * c], q[4 * c + 1], q[4 * c + 2], q[4 * c + 3]};
            *reinterpret_cast<f32x4*>(mine + 32 + 4 * c) = f32x4{dO[4 * c], dO[4 * c + 1], dO[4 * c + 2], dO[4 * c + 3]};
        }
        *reinterpret_cast<f32x4*>(mine + 64) = f32x4{m, inv_l, D, 0.f};
        float kj[32], vj[32], dk[32], dv[32];
        ksr_loadn<ST, 32>(a.qkv, n * 192 + 64 + 32 * h, kj);
        ksr_loadn<ST, 32>(a.qkv, n * 192 + 128 + 32 * h, vj);
#pragma unroll
        for (int i = 0; i < 32; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
        __syncthreads();
        for (int i = 0; i < S; ++i) {
            const float* qp = &xb[(i * PPW + pl) * XLD + 68 * h];
            float sc = 0.f, dp = 0.f;
            f32x4 qq[8], dd[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                qq[c] = *reinterpret_cast<const f32x4*>(qp + 4 * c);
                dd[c] = *reinterpret_cast<const f32x4*>(qp + 32 + 4 * c);
                sc += qq[c][0] * kj[4 * c] + qq[c][1] * kj[4 * c + 1] + qq[c][2] * kj[4 * c + 2] + qq[c][3] * kj[4 * c + 3];
                dp += dd[c][0] * vj[4 * c] + dd[c][1] * vj[4 * c + 1] + dd[c][2] * vj[4 * c + 2] + dd[c][3] * vj[4 * c + 3];
            }
            const f32x4 st = *reinterpret_cast<const f32x4*>(qp + 64);
            const float pij = expf(sc - st[0]) * st[1];
            float pt = pij;
            if (drop) {
                const long ni = (long)i * HW + p;          // global token index of query i
                const bool kept = ksr_attn_keep(ksr_attn_drop_base(a.seed, (unsigned long long)ni, h), b, keep);
                pt = kept ? pij * inv_keep : 0.f;
                dp = kept ? dp * inv_keep : 0.f;
            }
            const float dS = pij * (dp - st[2]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dk[4 * c + e] += dS * qq[c][e];
                    dv[4 * c + e] += pt * dd[c][e];
                }
            }
        }
        if (live) {
            ksr_storen<ST, 32>(a.dqkv, n * 192 + 64 + 32 * h, dk);
            ksr_storen<ST, 32>(a.dqkv, n * 192 + 128 + 32 * h, dv);
        }
    }
}

}  // namespace

// params[6]: norm_final.weight (or NULL), norm2.weight, norm1.weight, linear2.weight, linear1.weight,
// out_proj.weight.  saved[7]: xhat_final, rstd_final, xhat2, rstd2, hmid, xhat1, rstd1.
// outs[7]: ds2 (only with the final norm), dm, dpre [N,32], ds1, du1, da (only when p_drop > 0), dctx (NULL: the kernel
// stops at du1 and writes neither da nor dctx; ksr_batch_tx_layer_bwd_out_wg continues from du1).
// ln_slab (optional): 3 * 768 * 128 floats of workspace; with it the kernel also reduces the LayerNorm gain / shift
// gradients, ln_grads[l] (l = final norm, norm2, norm1; NULL to skip) receiving the gain gradient at [c] and the shift
// gradient at [ln_row_stride + c], and the outputs ds2 / ds1 (which only those reductions consume) may be NULL.
extern "C" int ksr_batch_tx_layer_bwd_token(const float* dy, const float* const* params, const float* const* saved,
                                            float* const* outs, int64_t N, float p_drop, uint64_t seed,
                                            float* ln_slab, float* const* ln_grads, int ln_row_stride,
                                            hipStream_t stream) {
    KSR_CHECK_ARG(dy && params && saved && outs && N > 0, "batch_tx_layer_bwd_token: bad argument");
    TxBwdTokArgs a{};
    a.dy = dy;
    a.nf_w = params[0]; a.n2_w = params[1]; a.n1_w = params[2]; a.l2_w = params[3]; a.l1_w = params[4]; a.out_w = params[5];
    a.xhatf = saved[0]; a.rstdf = saved[1]; a.xhat2 = saved[2]; a.rstd2 = saved[3]; a.hmid = saved[4];
    a.xhat1 = saved[5]; a.rstd1 = saved[6];
    a.ds2 = outs[0]; a.dm = outs[1]; a.dpre = outs[2]; a.ds1 = outs[3]; a.du1 = outs[4]; a.da = outs[5]; a.dctx = outs[6];
    for (int i = 1; i < 6; ++i) KSR_CHECK_ARG(params[i] != nullptr, "batch_tx_layer_bwd_token: params[%d] is null", i);
    for (int i = 2; i < 7; ++i) KSR_CHECK_ARG(saved[i] != nullptr, "batch_tx_layer_bwd_token: saved[%d] is null", i);
    KSR_CHECK_ARG(a.dm && a.dpre && a.du1, "batch_tx_layer_bwd_token: null output");
    KSR_CHECK_ARG(!a.nf_w || (a.xhatf && a.rstdf), "batch_tx_layer_bwd_token: final norm needs xhat/rstd");
    KSR_CHECK_ARG(ln_slab || (a.ds1 && (!a.nf_w || a.ds2)), "batch_tx_layer_bwd_token: ds1 / ds2 are required without ln_slab");
    KSR_CHECK_ARG(!ln_slab || (ln_grads && ln_row_stride >= 64), "batch_tx_layer_bwd_token: ln_slab needs ln_grads and a row stride >= 64");
    a.ln_slab = ln_slab;
    KSR_CHECK_ARG(!(p_drop > 0.f) || a.da || !a.dctx, "batch_tx_layer_bwd_token: dropout needs the da output");
    a.N = (long)N; a.p_drop = p_drop; a.seed = seed; a.seed_dev = ksr_seed_dev();
    const long ntiles = (N + 127) / 128;
    unsigned grid = 0;
#define KSR_CALL(P)                                                                  \
    do {                                                                             \
        grid = ksr_persistent_grid(tx_bwd_token_kernel<P>, ntiles, 768);             \
        hipLaunchKernelGGL(tx_bwd_token_kernel<P>, dim3(grid), dim3(256), 0, stream, a); \
    } while (0)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    KSR_LAUNCH_CHECK("batch_tx_layer_bwd_token");
    if (ln_slab) return ksr_ln_reduce(ln_slab, (int)grid, a.nf_w ? ln_grads[0] : nullptr, ln_grads[1], ln_grads[2], ln_row_stride, stream);
    return KSR_OK;
}

extern "C" int ksr_batch_tx_layer_bwd_attn(const float* qkv, const float* ctx, const float* dctx, float* dqkv, int S,
                                           int HW, float p_drop, uint64_t seed, hipStream_t stream) {
    KSR_CHECK_ARG(qkv && ctx && dctx && dqkv && S > 0 && HW > 0, "batch_tx_layer_bwd_attn: bad argument");
    KSR_UNSUPPORTED(S > 128, "batch_tx_layer_bwd_attn: local batch %d > 128 not supported", S);
    TxBwdAttnArgs a{qkv, ctx, dctx, dqkv, S, HW, p_drop, seed, ksr_seed_dev()};
    const int PPW = 128 / S;
    const int ntiles = (HW + PPW - 1) / PPW;
    if (ksr_prec() == KSR_BF16S) hipLaunchKernelGGL(tx_bwd_attn_kernel<__bf16>, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(tx_bwd_attn_kernel<float>, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("batch_tx_layer_bwd_attn");
    return KSR_OK;
}

// slab: [3][G][2][64] (block l belongs to out_l; a null out_l is skipped).  While the calling thread records its slab
// reductions (ksr_reduce_defer) these sums are recorded with them - ln_reduce_kernel and the recorded form add the rows in the
// same order, so the bits do not depend on the mode - and the slab must then stay untouched until the flush.
int ksr_ln_reduce(const float* slab, int G, float* out0, float* out1, float* out2, int row_stride, hipStream_t stream) {
    float* outs[3] = {out0, out1, out2};
    if (ksr_reduce_deferring() && G > 32) {
        for (int l = 0; l < 3; ++l)
            if (outs[l])
                if (int rc = ksr_slab_reduce_rows(slab + (long)l * G * 128, G, 2, 64, 128, outs[l], row_stride, stream)) return rc;
        return KSR_OK;
    }
    LnRedArgs r{slab, G, {out0, out1, out2}, row_stride};
    const int n = out2 ? 3 : (out1 ? 2 : 1);
    hipLaunchKernelGGL(ln_reduce_kernel, dim3(n, 4), dim3(1024), 0, stream, r);
    KSR_LAUNCH_CHECK("ln_reduce");
    return KSR_OK;
}

// bf16 storage: the first half of the token backward with linear2's weight gradient inside (see tx_bwd_mlp_wg_kernel).
// dy [N,64]: gradient of the layer's output; xhat2 [N,64], rstd2 [N], hmid [N,32]: saved by the forward.  Outputs du2 [N,64],
// dpre [N,32]; ln_grad: norm2's gain gradient at [c], shift gradient at [ln_row_stride + c]; dW2 [64,32], db2 [64].
// ln_slab: 768 * 128 floats, wg_slab: ksr_tx_bwd_wg_slab_floats() floats.
extern "C" int ksr_batch_tx_layer_bwd_mlp_wg(const float* dy, const float* xhat2, const float* rstd2, const float* hmid,
                                             const float* n2_w, const float* l2_w, float* du2, float* dpre, int64_t N,
                                             float p_drop, uint64_t seed, float* ln_slab, float* ln_grad, int ln_row_stride,
                                             float* wg_slab, float* dW2, float* db2, hipStream_t stream) {
    KSR_CHECK_ARG(dy && xhat2 && rstd2 && hmid && n2_w && l2_w && du2 && dpre && ln_slab && ln_grad && wg_slab && dW2 && db2 &&
                  N > 0 && ln_row_stride >= 64, "batch_tx_layer_bwd_mlp_wg: bad argument");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "batch_tx_layer_bwd_mlp_wg: bf16 storage mode only");
    TxBwdMlpArgs a{dy, xhat2, rstd2, n2_w, hmid, l2_w, du2, dpre, ln_slab, wg_slab, (long)N, p_drop, seed, ksr_seed_dev()};
    const long ntiles = (N + 127) / 128;
    const unsigned grid = ksr_persistent_grid(tx_bwd_mlp_wg_kernel, ntiles, KSR_WGF_MAX_GRID);
    hipLaunchKernelGGL(tx_bwd_mlp_wg_kernel, dim3(grid), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("batch_tx_layer_bwd_mlp_wg");
    const int rc = ksr_ln_reduce(ln_slab, (int)grid, ln_grad, nullptr, nullptr, ln_row_stride, stream);
    if (rc != KSR_OK) return rc;
    return ksr_slab_reduce(wg_slab, (int)grid, 64, 32, dW2, db2, stream);
}

// ... and the second half with linear1's weight gradient inside (tx_bwd_ln1_wg_kernel): du1 [N,64] from du2, dpre, xhat1,
// rstd1; ln_grad: norm1's pair; dW1 [32,64] is the product with xhat1 (the caller maps it through norm1's gain / shift:
// ksr_wgrad_affine_input), db1 [32].  ln_slab: 768 * 128 floats.
extern "C" int ksr_batch_tx_layer_bwd_ln1_wg(const float* du2, const float* dpre, const float* xhat1, const float* rstd1,
                                             const float* n1_w, const float* l1_w, float* du1, int64_t N, float* ln_slab,
                                             float* ln_grad, int ln_row_stride, float* wg_slab, float* dW1, float* db1,
                                             hipStream_t stream) {
    KSR_CHECK_ARG(du2 && dpre && xhat1 && rstd1 && n1_w && l1_w && du1 && ln_slab && ln_grad && wg_slab && dW1 && db1 && N > 0 && ln_row_stride >= 64,
                  "batch_tx_layer_bwd_ln1_wg: bad argument");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "batch_tx_layer_bwd_ln1_wg: bf16 storage mode only");
    TxBwdLn1Args a{du2, dpre, xhat1, rstd1, n1_w, l1_w, du1, ln_slab, wg_slab, (long)N};
    const long ntiles = (N + 127) / 128;
    const unsigned grid = ksr_persistent_grid(tx_bwd_ln1_wg_kernel, ntiles, KSR_WGF_MAX_GRID);
    hipLaunchKernelGGL(tx_bwd_ln1_wg_kernel, dim3(grid), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("batch_tx_layer_bwd_ln1_wg");
    if (int rc = ksr_ln_reduce(ln_slab, (int)grid, ln_grad, nullptr, nullptr, ln_row_stride, stream)) return rc;
    return ksr_slab_reduce(wg_slab, (int)grid, 32, 64, dW1, db1, stream);
}

// ---------------------------------------------------------------------------------------------
extern "C" int64_t ksr_tx_bwd_wg_slab_floats(void) { return (int64_t)KSR_WGF_MAX_GRID * KSR_WGF_ROW; }

// dctx [N,64] = mask(du1) * W_out (the attention branch's dropout mask regenerated from seed), dW [64][64] / db [64] of
// out_proj from mask(du1) and ctx.  wg_slab: ksr_tx_bwd_wg_slab_floats() floats.
extern "C" int ksr_batch_tx_layer_bwd_out_wg(const float* du1, const float* ctx, const float* out_w, float* dctx, int64_t N,
                                             float p_drop, uint64_t seed, float* wg_slab, float* dW, float* db,
                                             hipStream_t stream) {
    KSR_CHECK_ARG(du1 && ctx && out_w && dctx && wg_slab && dW && db && N > 0, "batch_tx_layer_bwd_out_wg: bad argument");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "batch_tx_layer_bwd_out_wg: bf16 storage mode only");
    TxBwdOutArgs a{du1, ctx, out_w, dctx, wg_slab, (long)N, p_drop, seed, ksr_seed_dev()};
    const long ntiles = (N + 127) / 128;
    const unsigned grid = ksr_persistent_grid(tx_bwd_out_wg_kernel, ntiles, KSR_WGF_MAX_GRID);
    hipLaunchKernelGGL(tx_bwd_out_wg_kernel, dim3(grid), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("batch_tx_layer_bwd_out_wg");
    return ksr_slab_reduce(wg_slab, (int)grid, 64, 64, dW, db, stream);
}
