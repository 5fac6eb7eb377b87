// Batch-axis attention on the matrix cores (used when the local batch S is large; the fused layer kernel
// in ksr_tx.hip keeps the simple VALU form, which is cheaper for small S).
//
// Tokens of one pixel occupy consecutive slots (slot = pl*Sp + b, Sp = S padded to a power of two <= 32 or
// to a multiple of 32), a wave owns 32 slots, a workgroup 128.  Everything is computed TRANSPOSED, with
// the QUERY on the MFMA column (= lane) axis:
//   scores^T[key][query] = K * Q^T      A = K rows from LDS,        B = this lane's q (registers)
//   ctx^T[dim][query]    = V^T * P^T    A = V^T from LDS,           B = this lane's probabilities
// The second product consumes the first one's accumulator registers directly as its B operand
// (register r of lane half h is key (r&3)+8(r>>2)+4h of the block: the k-order of the A operand is
// arranged to match), so scores and probabilities never leave the register file.  The softmax row
// statistics need one exchange between the two lanes of a query.
#include "ksr_attn.h"

extern "C" int ksr_pointwise_fwd(const float* in, const float* w, const float* bias, float* out, int n_out, int act, int B,
                                 int H, int W, hipStream_t stream);      // ksr_conv.hip

namespace {

constexpr int KLD = 68;      // K rows: 64 channels + 4 pad
constexpr int VLD = 132;     // V^T rows: 128 slots + 4 pad

struct __attribute__((aligned(16))) AttnSmem {
    float k[128 * KLD];
    float vt[64 * VLD];
};

template <int NW>   // key blocks per query: Sp / 32 (1 when Sp <= 32)
__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(AttnArgs a) {
    KSR_MIX_SEED(a);
    __shared__ AttnSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    const int PPW = 128 / Sp;
    const int ntiles = (HW + PPW - 1) / PPW;
    const int slot = wv * 32 + tok;
    const int pl = slot / Sp, b = slot - pl * Sp;
    const float scale = 0.17677669529663687f;
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const bool drop = a.p_drop > 0.f;
    const int kbase0 = (Sp >= 32) ? pl * Sp : wv * 32;       // first key slot of this lane's key blocks
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const SlotMap smap{Sp, lg, NW > 1 ? 1 : (Sp >= 32)};     // more than one block per query implies Sp >= 64

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p = tile * PPW + pl;
        const bool live = (b < S) && (p < HW);
        const long n = live ? ((long)b * HW + p) : 0;
        const float* row = a.qkv + n * 192;

        // q in "head-split" layout: registers 0..15 = head 0 dims 16h.., 16..31 = head 1 dims 16h..
        float q[32];
        ksr_load16(row + 16 * h, q);
        ksr_load16(row + 32 + 16 * h, q + 16);
#pragma unroll
        for (int i = 0; i < 32; ++i) q[i] *= scale;
        {
            float kk[32], vv[32];
            ksr_load32(row + 64 + 32 * h, kk);
            ksr_load32(row + 128 + 32 * h, vv);
            __syncthreads();                        // previous tile's readers are done
            float* kd = &sm.k[slot * KLD + 32 * h];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                *reinterpret_cast<f32x4*>(kd + 4 * c) = f32x4{kk[4 * c], kk[4 * c + 1], kk[4 * c + 2], kk[4 * c + 3]};
#pragma unroll
            for (int i = 0; i < 32; ++i) sm.vt[(32 * h + i) * VLD + slot] = vv[i];
        }
        __syncthreads();

        // ---- scores^T = K q^T ------------------------------------------------------------------
        f32x16 sc[2][NW];
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* kp = &sm.k[(kbase0 + 32 * kb + tok) * KLD + 32 * X + 16 * h];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(kp + 4 * s4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], q[16 * X + 4 * s4 + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], q[16 * X + 4 * s4 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], q[16 * X + 4 * s4 + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], q[16 * X + 4 * s4 + 3], acc, 0, 0, 0);
                }
                sc[X][kb] = acc;
            }

        // ---- masked softmax over the keys of this query's pixel (rows are split over the lane pair) ----
        float inv_l[2];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            float m = -INFINITY;
            const unsigned dbase = drop ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int same, kbb;
                    smap.decode(kbase0, kb, key_of_reg(r, h), pl, same, kbb);
                    const bool ok = same && (kbb < S);
                    sc[X][kb][r] = ok ? sc[X][kb][r] : -INFINITY;
                    m = fmaxf(m, sc[X][kb][r]);
                }
            m = ksr_pair_max(m);
            float l = 0.f;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pj = expf(sc[X][kb][r] - m);
                    l += pj;
                    if (drop) {
                        int same, kbb;
                        smap.decode(kbase0, kb, key_of_reg(r, h), pl, same, kbb);
                        pj = ksr_attn_keep(dbase, kbb, keep) ? pj * inv_keep : 0.f;
                    }
                    sc[X][kb][r] = pj;
                }
            l = ksr_pair_sum(l);
            inv_l[X] = 1.f / l;
        }

        // ---- ctx^T = V^T P^T -------------------------------------------------------------------
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
                const float* vp = &sm.vt[(32 * X + tok) * VLD + kbase0 + 32 * kb + 4 * h];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(vp + 8 * g);    // keys 8g+4h .. +3
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], sc[X][kb][4 * g + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], sc[X][kb][4 * g + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], sc[X][kb][4 * g + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], sc[X][kb][4 * g + 3], acc, 0, 0, 0);
                }
            }
            if (live) {
                // register r of lane half h is channel 32X + (r&3) + 8(r>>2) + 4h: four 16-byte groups
                float* cp = a.ctx + n * 64 + 32 * X + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(cp + 8 * g) =
                        make_float4(acc[4 * g] * inv_l[X], acc[4 * g + 1] * inv_l[X], acc[4 * g + 2] * inv_l[X],
                                    acc[4 * g + 3] * inv_l[X]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------ backward
constexpr int RLD = 36;      // row-major per-head tiles: 32 dims + 4 pad
constexpr int TLD = 132;     // transposed per-head tiles: 128 slots + 4 pad

struct __attribute__((aligned(16))) AttnBwdSmem {
    float rowA[128 * RLD];   // phase A: K rows      | phase B: Q rows (scaled)
    float rowB[128 * RLD];   // phase A: V rows      | phase B: dO rows
    float trA[32 * TLD];     // phase A: K^T         | phase B: Q^T (scaled)
    float trB[32 * TLD];     //                      | phase B: dO^T
    float stat[128 * 4];     // per query slot: m, 1/l, D, dropout base (bits)
};

// 16 MFMAs: acc[key/query row][lane column] += sum over the head's 32 dims of rows[rbase + rho][:] * x[:]
__device__ __forceinline__ void rows_times_x(const float* __restrict__ rows, int rbase, int tok, int h,
                                             const float (&x)[16], f32x16& acc) {
    const float* rp = rows + (rbase + tok) * RLD + 16 * h;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(rp + 4 * s4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[4 * s4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[4 * s4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[4 * s4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[4 * s4 + 3], acc, 0, 0, 0);
    }
}

// 16 MFMAs: acc[dim row][lane column] += sum over the 32 slots of a block of tr[dim][cbase + slot] * y[slot],
// y given in accumulator order (register r of lane half h <-> slot (r&3)+8(r>>2)+4h)
__device__ __forceinline__ void tr_times_acc(const float* __restrict__ tr, int cbase, int tok, int h,
                                             const f32x16& y, f32x16& acc) {
    const float* tp = tr + tok * TLD + cbase + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(tp + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], y[4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], y[4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], y[4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], y[4 * g + 3], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void store_acc_dims(float* __restrict__ dst, int h, const f32x16& acc) {
    // register r of lane half h is dim (r&3) + 8(r>>2) + 4h: four 16-byte groups
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + 8 * g + 4 * h) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
}

template <int NW, bool DROP>
__global__ void __launch_bounds__(256, 2) attn_bwd_kernel(AttnBwdArgs a) {
    KSR_MIX_SEED(a);
    __shared__ AttnBwdSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    const int PPW = 128 / Sp;
    const int ntiles = (HW + PPW - 1) / PPW;
    const int slot = wv * 32 + tok;
    const int pl = slot / Sp, b = slot - pl * Sp;
    const float scale = 0.17677669529663687f;
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    constexpr bool drop = DROP;
    const int base0 = (Sp >= 32) ? pl * Sp : wv * 32;     // first slot of this lane's key / query blocks
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const SlotMap smap{Sp, lg, NW > 1 ? 1 : (Sp >= 32)};     // more than one block per query implies Sp >= 64

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p = tile * PPW + pl;
        const bool live = (b < S) && (p < HW);
        const long n = live ? ((long)b * HW + p) : 0;
        const float* row = a.qkv + n * 192;

#pragma unroll 1
        for (int X = 0; X < 2; ++X) {
            float qx[16], kx[16], vx[16], dox[16];
            float D;
            ksr_load16(row + 32 * X + 16 * h, qx);
            ksr_load16(row + 64 + 32 * X + 16 * h, kx);
            ksr_load16(row + 128 + 32 * X + 16 * h, vx);
            ksr_load16(a.dctx + n * 64 + 32 * X + 16 * h, dox);
            {
                float ox[16];
                ksr_load16(a.ctx + n * 64 + 32 * X + 16 * h, ox);
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { qx[i] *= scale; d += dox[i] * ox[i]; }
                D = ksr_pair_sum(d);
            }
            // ---------------- phase A: this lane is a query; LDS holds K rows, V rows, K^T ----------------
            __syncthreads();
            {
                float* ra = &sm.rowA[slot * RLD + 16 * h];
                float* rb = &sm.rowB[slot * RLD + 16 * h];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    *reinterpret_cast<f32x4*>(ra + 4 * c) = f32x4{kx[4 * c], kx[4 * c + 1], kx[4 * c + 2], kx[4 * c + 3]};
                    *reinterpret_cast<f32x4*>(rb + 4 * c) = f32x4{vx[4 * c], vx[4 * c + 1], vx[4 * c + 2], vx[4 * c + 3]};
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) sm.trA[(16 * h + i) * TLD + slot] = kx[i];
            }
            __syncthreads();
            float m = -INFINITY, inv_l;
            const unsigned dbase = drop ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
            {
                f32x16 st[NW];
#pragma unroll
                for (int kb = 0; kb < NW; ++kb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
                    rows_times_x(sm.rowA, base0 + 32 * kb, tok, h, qx, st[kb]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int same, kbb;
                        smap.decode(base0, kb, key_of_reg(r, h), pl, same, kbb);
                        st[kb][r] = (same && (kbb < S)) ? st[kb][r] : -INFINITY;
                        m = fmaxf(m, st[kb][r]);
                    }
                }
                m = ksr_pair_max(m);
                float l = 0.f;
#pragma unroll
                for (int kb = 0; kb < NW; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { st[kb][r] = expf(st[kb][r] - m); l += st[kb][r]; }
                inv_l = 1.f / ksr_pair_sum(l);
                f32x16 dq;
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < NW; ++kb) {
                    f32x16 dp;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
                    rows_times_x(sm.rowB, base0 + 32 * kb, tok, h, dox, dp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float dpr = dp[r];
                        if (drop) {
                            int same, kbb;
                            smap.decode(base0, kb, key_of_reg(r, h), pl, same, kbb);
                            dpr = ksr_attn_keep(dbase, kbb, keep) ? dpr * inv_keep : 0.f;
                        }
                        dp[r] = st[kb][r] * inv_l * (dpr - D) * scale;      // dS (scaled for dq)
                    }
                    tr_times_acc(sm.trA, base0 + 32 * kb, tok, h, dp, dq);
                }
                if (live) store_acc_dims(a.dqkv + n * 192 + 32 * X, h, dq);
            }
            // ---------------- phase B: this lane is a key; LDS holds Q rows, dO rows, Q^T, dO^T, stats ----------------
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            {
                float* ra = &sm.rowA[slot * RLD + 16 * h];
                float* rb = &sm.rowB[slot * RLD + 16 * h];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    *reinterpret_cast<f32x4*>(ra + 4 * c) = f32x4{qx[4 * c], qx[4 * c + 1], qx[4 * c + 2], qx[4 * c + 3]};
                    *reinterpret_cast<f32x4*>(rb + 4 * c) = f32x4{dox[4 * c], dox[4 * c + 1], dox[4 * c + 2], dox[4 * c + 3]};
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sm.trA[(16 * h + i) * TLD + slot] = qx[i];
                    sm.trB[(16 * h + i) * TLD + slot] = dox[i];
                }
                if (h == 0) *reinterpret_cast<f32x4*>(&sm.stat[slot * 4]) = f32x4{m, inv_l, D, __uint_as_float(dbase)};
            }
            __syncthreads();
            {
                f32x16 dk, dv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
                for (int qb = 0; qb < NW; ++qb) {
                    f32x16 sq, dp;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { sq[r] = 0.f; dp[r] = 0.f; }
                    rows_times_x(sm.rowA, base0 + 32 * qb, tok, h, kx, sq);
                    rows_times_x(sm.rowB, base0 + 32 * qb, tok, h, vx, dp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qs = base0 + 32 * qb + key_of_reg(r, h);      // query slot
                        int same, qbb;
                        smap.decode(base0, qb, key_of_reg(r, h), pl, same, qbb);
                        const int qp = p;                                         // same pixel as this key
                        const bool ok = same && (qbb < S) && (qp < HW);
                        const f32x4 stt = *reinterpret_cast<const f32x4*>(&sm.stat[qs * 4]);
                        const float pij = ok ? expf(sq[r] - stt[0]) * stt[1] : 0.f;
                        float pt = pij, dpr = dp[r];
                        if (drop) {
                            const bool kept = ksr_attn_keep(__float_as_uint(stt[3]), b, keep);
                            pt = kept ? pij * inv_keep : 0.f;
                            dpr = kept ? dpr * inv_keep : 0.f;
                        }
                        sq[r] = pij * (dpr - stt[2]);    // dS
                        dp[r] = pt;                       // probabilities after dropout
                    }
                    tr_times_acc(sm.trA, base0 + 32 * qb, tok, h, sq, dk);
                    tr_times_acc(sm.trB, base0 + 32 * qb, tok, h, dp, dv);
                }
                if (live) {
                    store_acc_dims(a.dqkv + n * 192 + 64 + 32 * X, h, dk);
                    store_acc_dims(a.dqkv + n * 192 + 128 + 32 * X, h, dv);
                }
            }
        }
    }
}

// Slots per pixel: the next power of two (128 must be a multiple of it: a workgroup tile holds whole pixels).
int padded_S(int S) {
    int sp = 1;
    while (sp < S) sp *= 2;
    return sp;
}

}  // namespace

// ctx [N,64] = softmax(q k^T / sqrt(32)) v over the S images of each pixel, 2 heads; qkv [N,192] =
// in_proj(x) with rows (b, pixel).  p_drop > 0: dropout on the probabilities (same generator and
// indexing as the fused layer kernel).
extern "C" int ksr_batch_attn_fwd(const float* qkv, float* ctx, int S, int HW, float p_drop, uint64_t seed,
                                  hipStream_t stream) {
    KSR_CHECK_ARG(qkv && ctx && S > 0 && HW > 0, "batch_attn_fwd: bad argument");
    KSR_UNSUPPORTED(S > 128, "batch_attn_fwd: local batch %d > 128 not supported", S);
    KSR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "batch_attn_fwd: dropout p must be in [0,1)");
    AttnArgs a{qkv, ctx, S, padded_S(S), HW, p_drop, seed, ksr_seed_dev()};
    const int PPW = 128 / a.Sp;
    const int ntiles = (HW + PPW - 1) / PPW;
    const dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512)), block(256);
    if (ksr_prec() != KSR_F32) return ksr_attn_bf16_fwd_launch(a, a.Sp <= 32 ? 1 : a.Sp / 32, grid.x, stream);
    switch (a.Sp <= 32 ? 1 : a.Sp / 32) {
        case 1: hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, block, 0, stream, a); break;
    }
    KSR_LAUNCH_CHECK("batch_attn_fwd");
    return KSR_OK;
}

// in_proj + attention in one launch: qkv [N,192] = x [N,64] * in_w^T + in_b (written when qkv is not null: the
// backward's saved tensor) and ctx as ksr_batch_attn_fwd.  bf16 storage with 3 <= S <= 64 runs the fused kernel
// (ksr_attn_dma.hip); every other case is ksr_pointwise_fwd followed by ksr_batch_attn_fwd and needs the qkv buffer.
extern "C" int ksr_batch_qkv_attn_fwd(const float* x, const float* in_w, const float* in_b, float* qkv, float* ctx, int S,
                                      int HW, float p_drop, uint64_t seed, hipStream_t stream) {
    KSR_CHECK_ARG(x && in_w && in_b && ctx && S > 0 && HW > 0, "batch_qkv_attn_fwd: bad argument");
    KSR_UNSUPPORTED(S > 128, "batch_qkv_attn_fwd: local batch %d > 128 not supported", S);
    KSR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "batch_qkv_attn_fwd: dropout p must be in [0,1)");
    const int Sp = padded_S(S);
    if (ksr_prec() == KSR_BF16S && ksr_attn_dma_supported(Sp))
        return ksr_qkv_attn_dma_fwd_launch(x, in_w, in_b, qkv, ctx, S, Sp, HW, p_drop, seed, ksr_seed_dev(), stream);
    KSR_CHECK_ARG(qkv, "batch_qkv_attn_fwd: this precision / batch size needs the qkv buffer");
    if (int rc = ksr_pointwise_fwd(x, in_w, in_b, qkv, 192, 0, S, 1, HW, stream)) return rc;
    return ksr_batch_attn_fwd(qkv, ctx, S, HW, p_drop, seed, stream);
}

// Backward of ksr_batch_attn_fwd / of the attention inside ksr_batch_tx_layer_fwd: dqkv [N,192] from
// qkv, ctx (both saved by the forward) and dctx.  Same results as ksr_batch_tx_layer_bwd_attn, computed on
// the matrix cores.
extern "C" int ksr_batch_attn_bwd(const float* qkv, const float* ctx, const float* dctx, float* dqkv, int S, int HW,
                                  float p_drop, uint64_t seed, hipStream_t stream) {
    KSR_CHECK_ARG(qkv && ctx && dctx && dqkv && S > 0 && HW > 0, "batch_attn_bwd: bad argument");
    KSR_UNSUPPORTED(S > 128, "batch_attn_bwd: local batch %d > 128 not supported", S);
    AttnBwdArgs a{qkv, ctx, dctx, dqkv, S, padded_S(S), HW, p_drop, seed, ksr_seed_dev()};
    const int PPW = 128 / a.Sp;
    const int ntiles = (HW + PPW - 1) / PPW;
    const dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512)), block(256);
    if (ksr_prec() != KSR_F32) return ksr_attn_bf16_bwd_launch(a, a.Sp <= 32 ? 1 : a.Sp / 32, grid.x, stream);
    switch (a.Sp <= 32 ? 1 : a.Sp / 32) {
#define KSR_ATTN_BWD(NWV)                                                                       \
    do {                                                                                        \
        if (p_drop > 0.f) hipLaunchKernelGGL((attn_bwd_kernel<NWV, true>), grid, block, 0, stream, a);   \
        else hipLaunchKernelGGL((attn_bwd_kernel<NWV, false>), grid, block, 0, stream, a);              \
    } while (0)
        case 1: KSR_ATTN_BWD(1); break;
        case 2: KSR_ATTN_BWD(2); break;
        default: KSR_ATTN_BWD(4); break;
    }
#undef KSR_ATTN_BWD
    KSR_LAUNCH_CHECK("batch_attn_bwd");
    return KSR_OK;
}
