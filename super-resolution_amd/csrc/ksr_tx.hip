// Post-LN transformer encoder layer whose attention runs over the BATCH axis: the sequence at
// pixel p is the S = B_local feature vectors {x[b, p, :]}, E = 64, 2 heads of 32, FFN width 32.
//
// One lane owns one token (pixel, image) half: lane (tok, h) holds channels 32h..32h+31, which
// is exactly attention head h.  QKV / out-proj / MLP run on MFMA through ksr_wave_gemm with every
// weight staged once per (persistent) workgroup in LDS; the residual stream, LayerNorms, softmax and
// both heads stay in registers.  Only K and V cross lanes, through a padded LDS exchange buffer:
// lane (tok, h) walks the S keys of its own pixel and head with an online softmax.
//
// A workgroup handles 128 token slots = PPW pixels x S images (PPW = 128 / S); slot = b * PPW + pl so
// that neighbouring lanes read neighbouring pixels (contiguous 256-byte rows).
#include "ksr_gemm.h"

#include <cstddef>

namespace {

struct TxArgs {
    const float* in;    // [S, HW, 64]
    const float* ctx_in; // POST mode: attention output [N,64]
    float* out;         // [S, HW, 64]
    const float *in_w, *in_b;     // [192, 64], [192]
    const float *out_w, *out_b;   // [64, 64], [64]
    const float *l1_w, *l1_b;     // [32, 64], [32]
    const float *l2_w, *l2_b;     // [64, 32], [64]
    const float *n1_w, *n1_b, *n2_w, *n2_b;   // [64] each
    const float *nf_w, *nf_b;     // optional final LayerNorm applied to the output (may be null)
    // training-mode saves (all optional): q|k|v [N,192], ctx [N,64], xhat1 [N,64], rstd1 [N],
    // s1 [N,64], hmid [N,32], xhat2 [N,64], rstd2 [N]
    float *sv_qkv, *sv_ctx, *sv_xhat1, *sv_rstd1, *sv_s1, *sv_hmid, *sv_xhat2, *sv_rstd2;
    float *sv_xhatf, *sv_rstdf;   // final LayerNorm (only with nf_w)
    const float* in32;            // optional fp32 copy of `in` (read instead of it, whatever the storage mode)
    int S, HW;
    float eps;
    float p_drop;                 // 0 in eval mode
    unsigned long long seed;      // dropout stream id
    const unsigned long long* seed_dev;   // optional device word mixed into `seed` at kernel start (ksr_set_dropout_seed_source)
};

constexpr int KV_LD = 132;        // floats per slot: [h0: k32 v32][h1: k32 v32] + 4 pad

// MODE 0: whole layer (QKV projection, attention over S, rest).  MODE 1 ("post"): starts from ctx.
// MODE 2: one image per pixel (S = 1): softmax over a single key is 1, so ctx = V projection of the token and
// neither Q, K nor the exchange buffer exist; the out-projection then acts on a linear function of the token, and the
// kernel applies the product matrix W_out W_v (bias W_out b_v + b_out), which every workgroup forms once in LDS.
// Tiles are consecutive rows as in MODE 1.
// NW: waves per workgroup (4; 8 for the split-fp32 instance of MODE 2, whose weight planes leave room for one workgroup
// per CU).  X3: the weight arrays hold three bf16 planes (1.5 x the fp32 size).
template <int MODE, int NW = 4, bool X3 = false>
struct __attribute__((aligned(16))) TxSmem {
    static constexpr bool POST = MODE != 0;
    static constexpr int FS = X3 ? 3 : 2;         // weight bytes per element / 2
    float f_in[MODE == 0 ? 6 * 32 * 64 : 4];      // 64 -> 192
    float f_out[2 * 32 * 64 * FS / 2];     // 64 -> 64
    float f_l1[1 * 32 * 64 * FS / 2];      // 64 -> 32
    float f_l2[2 * 16 * 64 * FS / 2];      // 32 -> 64
    float b_in[6 * 32], b_out[2 * 32], b_l1[32], b_l2[2 * 32];
    float ln[6][64];              // n1_w, n1_b, n2_w, n2_b, nf_w, nf_b
    float kv[POST ? 4 : 128 * KV_LD];
    float xpose[POST ? NW * KSR_XPOSE_FLOATS : 4];   // POST: per-wave transposers of the coalesced loads / stores
};

__device__ __forceinline__ void ln_affine(const float (&xhat)[32], const float* __restrict__ g,
                                          const float* __restrict__ b, int h, float (&y)[32]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 gw = *reinterpret_cast<const f32x4*>(g + 32 * h + 4 * q);
        const f32x4 bw = *reinterpret_cast<const f32x4*>(b + 32 * h + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[4 * q + e] = xhat[4 * q + e] * gw[e] + bw[e];
    }
}

// MODE 1 (post): the QKV projection and the attention ran in separate kernels (ksr_pointwise_fwd,
// ksr_batch_attn_fwd); this kernel starts from ctx and needs neither the QKV weights nor the K/V
// exchange buffer in LDS (4x the occupancy).  MODE 2: see TxSmem.
template <int PREC, bool TRAIN, int MODE, int NW = 4>
__global__ void __launch_bounds__(64 * NW, MODE != 0 ? (NW + 3) / 4 * (NW > 4 ? 1 : 2) : 1) tx_layer_fwd_kernel(TxArgs a) {
    KSR_MIX_SEED(a);
    using ST = typename KsrStore<PREC>::T;
    constexpr bool POST = MODE != 0;          // row tiles, coalesced loads / stores, no cross-token step
    constexpr int NT = 64 * NW, TILE = 32 * NW;   // threads per workgroup, tokens per tile (row-tiled modes)
    static_assert(NW == 4 || MODE == 2, "only the single-image mode has a wider workgroup");
    __shared__ TxSmem<MODE, NW, PREC == KSR_F32X3> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;

    if (MODE == 0) ksr_stage_weights<PREC, 32, 6, false>(sm.f_in, sm.b_in, a.in_w, 64, a.in_b, MapHL{96}, MapHL{32}, tid, NT);
    if (MODE == 2) {
        // W_out and W_v (rows 128..191 of in_proj) -> the transposer area; their product -> the (not yet staged)
        // MLP fragment area, row-major; then staged like any weight matrix
        float* wo = sm.xpose;
        float* wvv = sm.xpose + 4096;
        float* prod = sm.f_l1;                       // f_l1 | f_l2 are adjacent: 4096 floats
        using SM = TxSmem<MODE, NW, PREC == KSR_F32X3>;
        static_assert(offsetof(SM, f_l2) == offsetof(SM, f_l1) + sizeof(sm.f_l1) && sizeof(sm.f_l1) + sizeof(sm.f_l2) >= 4096 * sizeof(float),
                      "f_l1 and f_l2 must be adjacent and hold the 64 x 64 product");
        for (int i = tid; i < 1024; i += NT) {
            reinterpret_cast<float4*>(wo)[i] = reinterpret_cast<const float4*>(a.out_w)[i];
            reinterpret_cast<float4*>(wvv)[i] = reinterpret_cast<const float4*>(a.in_w + 128 * 64)[i];
        }
        __syncthreads();
        if (tid < 256) {
            const int i = tid >> 2, j0 = (tid & 3) * 16;
            float acc[16], bsum = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            for (int k = 0; k < 64; ++k) {
                const float w = wo[i * 64 + k];
                bsum = fmaf(w, a.in_b[128 + k], bsum);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(wvv + k * 64 + j0 + 4 * q);
                    acc[4 * q] = fmaf(w, v.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w, v.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(w, v.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w, v.w, acc[4 * q + 3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) prod[i * 64 + j0 + j] = acc[j];
            if ((tid & 3) == 0) sm.b_in[i] = bsum + a.out_b[i];
        }
        __syncthreads();
        ksr_stage_weights<PREC, 32, 2, false>(sm.f_out, sm.b_out, prod, 64, sm.b_in, MapHL{32}, MapHL{32}, tid, NT);
        __syncthreads();
    } else {
        ksr_stage_weights<PREC, 32, 2, false>(sm.f_out, sm.b_out, a.out_w, 64, a.out_b, MapHL{32}, MapHL{32}, tid, NT);
    }
    ksr_stage_weights<PREC, 32, 1, false>(sm.f_l1, sm.b_l1, a.l1_w, 64, a.l1_b, MapSplit{16}, MapHL{32}, tid, NT);
    ksr_stage_weights<PREC, 16, 2, false>(sm.f_l2, sm.b_l2, a.l2_w, 32, a.l2_b, MapHL{32}, MapSplit{16}, tid, NT);
    if (tid < 64) {
        sm.ln[0][tid] = a.n1_w[tid]; sm.ln[1][tid] = a.n1_b[tid];
        sm.ln[2][tid] = a.n2_w[tid]; sm.ln[3][tid] = a.n2_b[tid];
        sm.ln[4][tid] = a.nf_w ? a.nf_w[tid] : 1.f; sm.ln[5][tid] = a.nf_b ? a.nf_b[tid] : 0.f;
    }
    __syncthreads();

    const int S = a.S, HW = a.HW;
    const int PPW = POST ? TILE : 128 / S;            // pixels per workgroup tile (S <= 128)
    const int ntiles = POST ? (int)(((long)S * HW + TILE - 1) / TILE) : (HW + PPW - 1) / PPW;
    const int slot = wv * 32 + tok;
    const int b = slot / PPW, pl = slot - b * PPW;
    const float scale = 0.17677669529663687f;         // 1/sqrt(32)
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;

    // row-tiled modes: the next tile's rows are requested before this tile's arithmetic (bf16 storage, or the
    // single-image mode whose only input is one tile)
    constexpr bool AHEAD = (MODE == 2) || (MODE == 1 && sizeof(ST) == 2);
    KsrTile<ST> nx_in, nx_ctx;
    auto fetch_next = [&](int t) {
        const long r0 = (long)t * TILE + wv * 32, left = (long)S * HW - r0;
        const int nv = left < 32 ? (int)(left > 0 ? left : 0) : 32;
        if (!a.in32) ksr_co_fetch<ST, true>(a.in, r0, nv, 64, 0, lane, nx_in);
        if (MODE == 1) ksr_co_fetch<ST, true>(a.ctx_in, r0, nv, 64, 0, lane, nx_ctx);
    };
    if (AHEAD && (int)blockIdx.x < ntiles) fetch_next(blockIdx.x);
    // single-image mode: a tile's output is stored during the next tile (ksr_pend32_co)
    constexpr bool DEFER = MODE == 2;
    KsrPend<ST> pend;
    pend.nvalid = 0; pend.n0 = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // POST: no cross-token step is left, tiles are 128 consecutive rows of [S*HW, 64]
        const int p = tile * PPW + pl;
        const long nrow = (long)tile * TILE + slot;
        const bool live = POST ? (nrow < (long)S * HW) : ((b < S) && (p < HW));
        const long n = live ? (POST ? nrow : ((long)b * HW + p)) : 0;

        // POST: a wave's 32 rows are consecutive, loads and stores go through the LDS transposer (coalesced)
        const long n0w = (long)tile * TILE + wv * 32;
        const int nvalid = POST ? (((long)S * HW - n0w) < 32 ? (int)((long)S * HW - n0w > 0 ? (long)S * HW - n0w : 0) : 32) : 0;
        float* xb = sm.xpose + (POST ? wv * KSR_XPOSE_FLOATS : 0);
        auto st32 = [&](float* base, const float (&v)[32]) {
            if (POST) ksr_store32_co<ST, !AHEAD>(xb, base, n0w, nvalid, 64, 0, lane, v);
            else if (live) ksr_storen<ST, 32>(base, n * 64 + 32 * h, v);
        };
        // tensors kept for the backward: non-temporal stores in the row-tiled mode
        auto sv32 = [&](float* base, const float (&v)[32]) {
            if (POST) ksr_store32_co<ST, !AHEAD, true>(xb, base, n0w, nvalid, 64, 0, lane, v);
            else if (live) ksr_storen<ST, 32>(base, n * 64 + 32 * h, v);
        };

        float s0[32];
        if (POST) {
            if (a.in32) ksr_load32_co<float>(xb, a.in32, n0w, nvalid, 64, 0, lane, s0);
            else if (AHEAD) ksr_co_finish<ST>(xb, lane, nx_in, s0);
            else ksr_load32_co<ST>(xb, a.in, n0w, nvalid, 64, 0, lane, s0);
        } else if (live) {
            if (a.in32) ksr_loadn<float, 32>(a.in32, n * 64 + 32 * h, s0);
            else ksr_loadn<ST, 32>(a.in, n * 64 + 32 * h, s0);
        }
        else {
#pragma unroll
            for (int i = 0; i < 32; ++i) s0[i] = 0.f;
        }

        float ctx[32];
        if (MODE == 1) {
            if (AHEAD) ksr_co_finish<ST>(xb, lane, nx_ctx, ctx);
            else ksr_load32_co<ST>(xb, a.ctx_in, n0w, nvalid, 64, 0, lane, ctx);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) ctx[i] = s0[i];      // f_out holds W_out W_v
        } else {
        // ---- QKV projection -------------------------------------------------------------
        float q[32];
        {
            f32x16 acc[6];
            ksr_wave_gemm<PREC, 32, 6>(sm.f_in, sm.b_in, s0, acc, lane);
            float kvr[64];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                q[i] = acc[0][i]; q[16 + i] = acc[1][i];
                kvr[i] = acc[2][i]; kvr[16 + i] = acc[3][i];
                kvr[32 + i] = acc[4][i]; kvr[48 + i] = acc[5][i];
            }
            if (TRAIN && a.sv_qkv && live) {
                ksr_storen<ST, 32>(a.sv_qkv, n * 192 + 32 * h, q);
                ksr_storen<ST, 32>(a.sv_qkv, n * 192 + 64 + 32 * h, kvr);
                ksr_storen<ST, 32>(a.sv_qkv, n * 192 + 128 + 32 * h, kvr + 32);
            }
            __syncthreads();                          // previous tile's K/V readers are done
            float* dst = &sm.kv[slot * KV_LD + 64 * h];
#pragma unroll
            for (int c = 0; c < 16; ++c)
                *reinterpret_cast<f32x4*>(dst + 4 * c) =
                    f32x4{kvr[4 * c], kvr[4 * c + 1], kvr[4 * c + 2], kvr[4 * c + 3]};
#pragma unroll
            for (int i = 0; i < 32; ++i) q[i] *= scale;
        }
        __syncthreads();

        // ---- attention over the S images at this pixel, head h --------------------------
        {
            float m = -INFINITY, l = 0.f;
            const unsigned dbase = ksr_attn_drop_base(a.seed, (unsigned long long)n, h);
#pragma unroll
            for (int i = 0; i < 32; ++i) ctx[i] = 0.f;
            for (int j = 0; j < S; ++j) {
                const float* kp = &sm.kv[(j * PPW + pl) * KV_LD + 64 * h];
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 4 * c);
                    sc += q[4 * c] * kk[0] + q[4 * c + 1] * kk[1] + q[4 * c + 2] * kk[2] + q[4 * c + 3] * kk[3];
                }
                const float mn = fmaxf(m, sc);
                const float alpha = expf(m - mn);
                float pj = expf(sc - mn);
                l = l * alpha + pj;
                if (TRAIN && a.p_drop > 0.f) {
                    // dropout on the attention probabilities (after normalisation in the reference;
                    // the mask commutes with the 1/l scaling)
                    pj = ksr_attn_keep(dbase, j, keep) ? pj * inv_keep : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(kp + 32 + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ctx[4 * c + e] = ctx[4 * c + e] * alpha + pj * vv[e];
                }
                m = mn;
            }
            const float inv_l = 1.f / l;
#pragma unroll
            for (int i = 0; i < 32; ++i) ctx[i] *= inv_l;
            if (TRAIN && a.sv_ctx && live) ksr_storen<ST, 32>(a.sv_ctx, n * 64 + 32 * h, ctx);
        }

        }
        if (AHEAD && tile + (int)gridDim.x < ntiles) fetch_next(tile + gridDim.x);
        if (DEFER) ksr_pend_store<ST>(a.out, 64, 0, lane, pend);
        // ---- out projection + residual + LayerNorm 1 -------------------------------------
        float s1[32];
        {
            f32x16 acc[2];
            ksr_wave_gemm<PREC, 32, 2>(sm.f_out, sm.b_out, ctx, acc, lane);
            float u[32], xhat[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) { u[i] = acc[0][i]; u[16 + i] = acc[1][i]; }
            if (TRAIN && a.p_drop > 0.f) {
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    u[i] = ksr_keep(a.seed, 1u, (unsigned long long)n * 64 + 32 * h, i, keep) ? u[i] * inv_keep : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) u[i] += s0[i];
            const float rstd = ksr_layernorm64(u, xhat, a.eps);
            ln_affine(xhat, sm.ln[0], sm.ln[1], h, s1);
            if (TRAIN) {
                if (a.sv_xhat1) sv32(a.sv_xhat1, xhat);
                if (a.sv_s1) sv32(a.sv_s1, s1);
                if (live && a.sv_rstd1 && h == 0) a.sv_rstd1[n] = rstd;
            }
        }

        // ---- MLP 64 -> 32 -> 64 + residual + LayerNorm 2 ----------------------------------
        float s2[32];
        {
            f32x16 acc1[1];
            ksr_wave_gemm<PREC, 32, 1>(sm.f_l1, sm.b_l1, s1, acc1, lane);
            float hm[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) hm[i] = acc1[0][i] > 0.f ? acc1[0][i] : 0.f;
            if (TRAIN && a.p_drop > 0.f) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    hm[i] = ksr_keep(a.seed, 2u, (unsigned long long)n * 32 + 16 * h, i, keep) ? hm[i] * inv_keep : 0.f;
            }
            if (TRAIN && a.sv_hmid) {
                if (POST) ksr_store16_co<ST, !AHEAD, true>(xb, a.sv_hmid, n0w, nvalid, lane, hm);
                else if (live) ksr_storen<ST, 16>(a.sv_hmid, n * 32 + 16 * h, hm);
            }
            f32x16 acc2[2];
            ksr_wave_gemm<PREC, 16, 2>(sm.f_l2, sm.b_l2, hm, acc2, lane);
            float u[32], xhat[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) { u[i] = acc2[0][i]; u[16 + i] = acc2[1][i]; }
            if (TRAIN && a.p_drop > 0.f) {
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    u[i] = ksr_keep(a.seed, 3u, (unsigned long long)n * 64 + 32 * h, i, keep) ? u[i] * inv_keep : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) u[i] += s1[i];
            const float rstd = ksr_layernorm64(u, xhat, a.eps);
            ln_affine(xhat, sm.ln[2], sm.ln[3], h, s2);
            if (TRAIN) {
                if (a.sv_xhat2) sv32(a.sv_xhat2, xhat);
                if (live && a.sv_rstd2 && h == 0) a.sv_rstd2[n] = rstd;
            }
        }

        // ---- optional final LayerNorm of the encoder stack --------------------------------
        if (a.nf_w) {
            float xhat[32], y[32];
            const float rstd = ksr_layernorm64(s2, xhat, a.eps);
            ln_affine(xhat, sm.ln[4], sm.ln[5], h, y);
            if (TRAIN) {
                if (a.sv_xhatf) sv32(a.sv_xhatf, xhat);
                if (live && a.sv_rstdf && h == 0) a.sv_rstdf[n] = rstd;
            }
            if (DEFER) ksr_pend32_co<ST>(xb, n0w, nvalid, lane, y, pend); else st32(a.out, y);
        } else {
            if (DEFER) ksr_pend32_co<ST>(xb, n0w, nvalid, lane, s2, pend); else st32(a.out, s2);
        }
    }
    if (DEFER) ksr_pend_store<ST>(a.out, 64, 0, lane, pend);
}

// ---------------------------------------------------------------------------------------------
// Inference on one image: BOTH encoder layers and the final LayerNorm in one pass over the tokens (each layer as in MODE 2
// of tx_layer_fwd_kernel: W_out W_v applied as one matrix).  The [N,64] tensor between the layers (a 133-MB write and read
// per 540 x 960 frame) and one pair of transposes never exist.  Eight waves share one copy of both layers' weights
// (fp32: 64 KB; split fp32: three bf16 planes, 96 KB) beside half-width transposers: one workgroup per CU.
struct Tx2Args {
    const float* in; float* out;
    const float* p[2][12];        // per layer: the 12 tensors of ksr_batch_tx_layer_fwd's params
    const float *nf_w, *nf_b;
    long N; float eps;
};
template <int PREC>
struct __attribute__((aligned(16))) Tx2Smem {
    float f_out[2][ksr_frag_floats_p<PREC>(32, 2)];
    float f_l1[2][ksr_frag_floats_p<PREC>(32, 1)];      // f_l1[l] | f_l2[l] are not adjacent across layers: the product
    float f_l2[2][ksr_frag_floats_p<PREC>(16, 2)];      // matrix of layer l is formed in scratch below
    float b_out[2][64], b_l1[2][32], b_l2[2][64], b_v[64];
    float ln[2][4][64], lnf[2][64];
    float xpose[8 * KSR_XPOSE_HALF_FLOATS];             // 36 KB; the prologue uses 32 KB of it for W_out and W_v
    float prod[64 * 64];
};

template <int PREC>
__global__ void __launch_bounds__(512, 2) tx2_single_fwd_kernel(Tx2Args a) {
    __shared__ Tx2Smem<PREC> sm;
    static_assert(sizeof(sm.xpose) >= 2 * 4096 * sizeof(float), "the transposers hold W_out and W_v in the prologue");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = lane >> 5;
    for (int l = 0; l < 2; ++l) {
        const float *in_w = a.p[l][0], *in_b = a.p[l][1], *out_w = a.p[l][2], *out_b = a.p[l][3];
        float* wo = sm.xpose;
        float* wvv = sm.xpose + 4096;
        __syncthreads();
        for (int i = tid; i < 1024; i += 512) {
            reinterpret_cast<float4*>(wo)[i] = reinterpret_cast<const float4*>(out_w)[i];
            reinterpret_cast<float4*>(wvv)[i] = reinterpret_cast<const float4*>(in_w + 128 * 64)[i];
        }
        __syncthreads();
        if (tid < 256) {
            const int i = tid >> 2, j0 = (tid & 3) * 16;
            float acc[16], bsum = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            for (int k = 0; k < 64; ++k) {
                const float w = wo[i * 64 + k];
                bsum = fmaf(w, in_b[128 + k], bsum);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(wvv + k * 64 + j0 + 4 * q);
                    acc[4 * q] = fmaf(w, v.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w, v.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(w, v.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w, v.w, acc[4 * q + 3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) sm.prod[i * 64 + j0 + j] = acc[j];
            if ((tid & 3) == 0) sm.b_v[i] = bsum + out_b[i];
        }
        __syncthreads();
        ksr_stage_weights<PREC, 32, 2, false>(sm.f_out[l], sm.b_out[l], sm.prod, 64, sm.b_v, MapHL{32}, MapHL{32}, tid, 512);
        ksr_stage_weights<PREC, 32, 1, false>(sm.f_l1[l], sm.b_l1[l], a.p[l][4], 64, a.p[l][5], MapSplit{16}, MapHL{32}, tid, 512);
        ksr_stage_weights<PREC, 16, 2, false>(sm.f_l2[l], sm.b_l2[l], a.p[l][6], 32, a.p[l][7], MapHL{32}, MapSplit{16}, tid, 512);
        if (tid < 64) {
            sm.ln[l][0][tid] = a.p[l][8][tid]; sm.ln[l][1][tid] = a.p[l][9][tid];
            sm.ln[l][2][tid] = a.p[l][10][tid]; sm.ln[l][3][tid] = a.p[l][11][tid];
            if (l == 1) { sm.lnf[0][tid] = a.nf_w[tid]; sm.lnf[1][tid] = a.nf_b[tid]; }
        }
    }
    __syncthreads();

    const long ntiles = (a.N + 255) / 256;
    float* xb = sm.xpose + wv * KSR_XPOSE_HALF_FLOATS;
    KsrTile<float> nx;
    auto fetch_next = [&](long t) {
        const long r0 = t * 256 + wv * 32, left = a.N - r0;
        const int nv = left < 32 ? (int)(left > 0 ? left : 0) : 32;
        ksr_co_fetch<float, true>(a.in, r0, nv, 64, 0, lane, nx);
    };
    if ((long)blockIdx.x < ntiles) fetch_next(blockIdx.x);
    KsrPendHalf pend;
    pend.nvalid = 0; pend.n0 = 0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0w = tile * 256 + wv * 32;
        const int nvalid = (a.N - n0w) < 32 ? (int)(a.N - n0w > 0 ? a.N - n0w : 0) : 32;
        float s0[32];
        ksr_co_finish_half(xb, lane, nx, s0);
        if (tile + gridDim.x < ntiles) fetch_next(tile + gridDim.x);
        ksr_pend_store_half(a.out, 64, 0, lane, pend);          // the previous tile's output
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            float s1[32];
            {
                f32x16 acc[2];
                ksr_wave_gemm<PREC, 32, 2>(sm.f_out[l], sm.b_out[l], s0, acc, lane);
                float u[32], xhat[32];
#pragma unroll
                for (int i = 0; i < 16; ++i) { u[i] = acc[0][i] + s0[i]; u[16 + i] = acc[1][i] + s0[16 + i]; }
                ksr_layernorm64(u, xhat, a.eps);
                ln_affine(xhat, sm.ln[l][0], sm.ln[l][1], h, s1);
            }
            {
                f32x16 acc1[1];
                ksr_wave_gemm<PREC, 32, 1>(sm.f_l1[l], sm.b_l1[l], s1, acc1, lane);
                float hm[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) hm[i] = acc1[0][i] > 0.f ? acc1[0][i] : 0.f;
                f32x16 acc2[2];
                ksr_wave_gemm<PREC, 16, 2>(sm.f_l2[l], sm.b_l2[l], hm, acc2, lane);
                float u[32], xhat[32];
#pragma unroll
                for (int i = 0; i < 16; ++i) { u[i] = acc2[0][i] + s1[i]; u[16 + i] = acc2[1][i] + s1[16 + i]; }
                ksr_layernorm64(u, xhat, a.eps);
                ln_affine(xhat, sm.ln[l][2], sm.ln[l][3], h, s0);      // the next layer's input
            }
        }
        float xhat[32], y[32];
        ksr_layernorm64(s0, xhat, a.eps);
        ln_affine(xhat, sm.lnf[0], sm.lnf[1], h, y);
        ksr_pend32_co_half(xb, n0w, nvalid, lane, y, pend);
    }
    ksr_pend_store_half(a.out, 64, 0, lane, pend);
}

}  // namespace

extern "C" int ksr_batch_tx_layer_fwd(const float* in, float* out, const float* const* params, float* const* saves,
                                      int S, int HW, float eps, float p_drop, uint64_t seed, const float* in_f32_copy,
                                      hipStream_t stream) {
    KSR_CHECK_ARG(in && out && params, "batch_tx_layer_fwd: null pointer");
    KSR_CHECK_ARG(S > 0 && HW > 0, "batch_tx_layer_fwd: non-positive dimension");
    KSR_UNSUPPORTED(S > 128, "batch_tx_layer_fwd: local batch %d > 128 not supported", S);
    KSR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "batch_tx_layer_fwd: dropout p must be in [0,1)");
    for (int i = 0; i < 12; ++i) KSR_CHECK_ARG(params[i] != nullptr, "batch_tx_layer_fwd: params[%d] is null", i);
    TxArgs a{};
    a.in = in; a.in32 = in_f32_copy; a.out = out;
    a.in_w = params[0]; a.in_b = params[1]; a.out_w = params[2]; a.out_b = params[3];
    a.l1_w = params[4]; a.l1_b = params[5]; a.l2_w = params[6]; a.l2_b = params[7];
    a.n1_w = params[8]; a.n1_b = params[9]; a.n2_w = params[10]; a.n2_b = params[11];
    a.nf_w = params[12]; a.nf_b = params[13];
    KSR_CHECK_ARG((a.nf_w == nullptr) == (a.nf_b == nullptr), "batch_tx_layer_fwd: final norm needs weight and bias");
    if (saves) {
        a.sv_qkv = saves[0]; a.sv_ctx = saves[1]; a.sv_xhat1 = saves[2]; a.sv_rstd1 = saves[3]; a.sv_s1 = saves[4];
        a.sv_hmid = saves[5]; a.sv_xhat2 = saves[6]; a.sv_rstd2 = saves[7]; a.sv_xhatf = saves[8]; a.sv_rstdf = saves[9];
    }
    a.S = S; a.HW = HW; a.eps = eps; a.p_drop = p_drop; a.seed = seed; a.seed_dev = ksr_seed_dev();
    const int PPW = 128 / S;
    const int ntiles = (HW + PPW - 1) / PPW;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    const bool train = saves || p_drop > 0.f;
    if (S == 1 && !train) {      // inference on one image: no attention step is left (see TxSmem)
        // one resident round: every workgroup stages the weights (and forms W_out W_v) once
        if (ksr_prec() == KSR_F32 && ksr_f32_split()) {
            // fp32 products as six bf16 matrix instructions: eight waves share one copy of the three weight planes
            constexpr int XW = 8;
            const long nt8 = ((long)HW + 32 * XW - 1) / (32 * XW);
            hipLaunchKernelGGL((tx_layer_fwd_kernel<KSR_F32X3, false, 2, XW>), dim3(ksr_persistent_grid(tx_layer_fwd_kernel<KSR_F32X3, false, 2, XW>, nt8, 1024, 64 * XW)),
                               dim3(64 * XW), 0, stream, a);
            KSR_LAUNCH_CHECK("batch_tx_layer_fwd(single image, split fp32)");
            return KSR_OK;
        }
#define KSR_CALL(P) hipLaunchKernelGGL((tx_layer_fwd_kernel<P, false, 2>), dim3(ksr_persistent_grid(tx_layer_fwd_kernel<P, false, 2>, ntiles, 1024)), dim3(256), 0, stream, a)
        KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
        KSR_LAUNCH_CHECK("batch_tx_layer_fwd(single image)");
        return KSR_OK;
    }
#define KSR_CALL(P)                                                                                                   \
    do {                                                                                                              \
        if (train) hipLaunchKernelGGL((tx_layer_fwd_kernel<P, true, 0>), dim3(grid), dim3(256), 0, stream, a);    \
        else hipLaunchKernelGGL((tx_layer_fwd_kernel<P, false, 0>), dim3(grid), dim3(256), 0, stream, a);         \
    } while (0)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    KSR_LAUNCH_CHECK("batch_tx_layer_fwd");
    return KSR_OK;
}

// The same layer after its attention: in = layer input (residual), ctx = attention output
// (ksr_batch_attn_fwd).  params as ksr_batch_tx_layer_fwd (entries 0, 1 unused); saves[0], saves[1] unused.
extern "C" int ksr_batch_tx_layer_post_fwd(const float* in, const float* ctx, float* out, const float* const* params,
                                           float* const* saves, int64_t N, float eps, float p_drop, uint64_t seed,
                                           const float* in_f32_copy, hipStream_t stream) {
    KSR_CHECK_ARG(in && ctx && out && params && N > 0, "batch_tx_layer_post_fwd: bad argument");
    KSR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "batch_tx_layer_post_fwd: dropout p must be in [0,1)");
    for (int i = 2; i < 12; ++i) KSR_CHECK_ARG(params[i] != nullptr, "batch_tx_layer_post_fwd: params[%d] is null", i);
    TxArgs a{};
    a.in = in; a.in32 = in_f32_copy; a.ctx_in = ctx; a.out = out;
    a.out_w = params[2]; a.out_b = params[3];
    a.l1_w = params[4]; a.l1_b = params[5]; a.l2_w = params[6]; a.l2_b = params[7];
    a.n1_w = params[8]; a.n1_b = params[9]; a.n2_w = params[10]; a.n2_b = params[11];
    a.nf_w = params[12]; a.nf_b = params[13];
    KSR_CHECK_ARG((a.nf_w == nullptr) == (a.nf_b == nullptr), "batch_tx_layer_post_fwd: final norm needs weight and bias");
    if (saves) {
        a.sv_xhat1 = saves[2]; a.sv_rstd1 = saves[3]; a.sv_s1 = saves[4];
        a.sv_hmid = saves[5]; a.sv_xhat2 = saves[6]; a.sv_rstd2 = saves[7]; a.sv_xhatf = saves[8]; a.sv_rstdf = saves[9];
    }
    a.S = 1; a.HW = (int)N; a.eps = eps; a.p_drop = p_drop; a.seed = seed; a.seed_dev = ksr_seed_dev();
    const long ntiles = (N + 127) / 128;
    const bool train = saves || p_drop > 0.f;
#define KSR_CALL(P)                                                                                                   \
    do {                                                                                                              \
        if (train) hipLaunchKernelGGL((tx_layer_fwd_kernel<P, true, 1>), dim3(ksr_persistent_grid(tx_layer_fwd_kernel<P, true, 1>, ntiles, 1024)), dim3(256), 0, stream, a);     \
        else hipLaunchKernelGGL((tx_layer_fwd_kernel<P, false, 1>), dim3(ksr_persistent_grid(tx_layer_fwd_kernel<P, false, 1>, ntiles, 1024)), dim3(256), 0, stream, a);          \
    } while (0)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    KSR_LAUNCH_CHECK("batch_tx_layer_post_fwd");
    return KSR_OK;
}

// Both encoder layers and the final LayerNorm of a single image in one launch (GazeTR.forward in eval mode on one image,
// gazetr.py:110-117: over a batch of one the attention is the V projection).  params0 / params1: the 12 tensors of
// ksr_batch_tx_layer_fwd per layer; in, out [N,64] fp32.  fp32 mode only (other modes: KSR_ERR_UNSUPPORTED).
extern "C" int ksr_tx2_single_fwd(const float* in, float* out, const float* const* params0, const float* const* params1,
                                  const float* nf_w, const float* nf_b, int64_t N, float eps, hipStream_t stream) {
    KSR_CHECK_ARG(in && out && params0 && params1 && nf_w && nf_b && N > 0, "tx2_single_fwd: bad argument");
    KSR_UNSUPPORTED(ksr_prec() != KSR_F32, "tx2_single_fwd: fp32 mode only");
    Tx2Args a{};
    a.in = in; a.out = out; a.nf_w = nf_w; a.nf_b = nf_b; a.N = (long)N; a.eps = eps;
    for (int i = 0; i < 12; ++i) {
        KSR_CHECK_ARG(params0[i] && params1[i], "tx2_single_fwd: params[%d] is null", i);
        a.p[0][i] = params0[i]; a.p[1][i] = params1[i];
    }
    const long ntiles = ((long)N + 255) / 256;
    if (ksr_f32_split())
        hipLaunchKernelGGL(tx2_single_fwd_kernel<KSR_F32X3>, dim3(ksr_persistent_grid(tx2_single_fwd_kernel<KSR_F32X3>, ntiles, 1024, 512)), dim3(512), 0, stream, a);
    else
        hipLaunchKernelGGL(tx2_single_fwd_kernel<KSR_F32>, dim3(ksr_persistent_grid(tx2_single_fwd_kernel<KSR_F32>, ntiles, 1024, 512)), dim3(512), 0, stream, a);
    KSR_LAUNCH_CHECK("tx2_single_fwd");
    return KSR_OK;
}
