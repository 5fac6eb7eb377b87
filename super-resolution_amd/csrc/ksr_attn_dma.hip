// Batch-axis attention for bf16 STORAGE (mode KSR_BF16S), 4 <= Sp <= 64: the operand rows go from global memory
// straight into LDS with the LDS-DMA (global_load_lds, 16 bytes per lane), one pass ahead of the arithmetic, and are
// never held in registers as fetched tiles.  Same products, masks, softmax and dropout arithmetic as ksr_attn_bf16.hip;
// what differs:
//   * a pass is 64 slots (64 / Sp pixels) x both heads: wave w works on head w & 1 and on the 32-slot block w >> 1, so a
//     staged row is a token's whole 128-byte q, k, v or dO row and every transfer instruction covers 8 full rows;
//   * two stages (2 x 32 KB backward, 2 x 24 KB forward) are separate __shared__ arrays: the compiler's wait-count
//     bookkeeping then knows that reading one does not depend on the transfers into the other.  Workgroup barriers are
//     issued as `s_waitcnt ... ; s_barrier` directly: __syncthreads() carries a fence that would also wait for the
//     transfers of the NEXT pass;
//   * a staged tensor is laid out as [8-row group][head][row & 7][4 chunks of 8 bf16], chunk c of row r at chunk position
//     c ^ ((r >> 2) & 3): the bank pattern of ksr_attn_bf16.hip's rows (conflict-free row reads and transposing reads).
//     The permutation is applied on the GLOBAL side of the transfer (the lane that fills position cp fetches chunk
//     cp ^ ((r >> 2) & 3)), the LDS side of a transfer instruction being lane-linear by construction;
//   * the softmax scale multiplies the scores inside the exponent (exp2(s * scale * log2 e - m')) and the key gradient
//     at its store, instead of being rounded into q;
//   * backward: D = rowsum(dO * O) is formed as sum_j P_ij dP~_ij (dP~ = dP after dropout) from the fp32 accumulators
//     that phase A holds anyway, so the forward's output is not read.
#include "ksr_attn.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TS = 64;                  // slots per pass
constexpr int TEN = TS * 64;            // elements of one staged tensor (64 rows x 64 channels, 8 KB)
constexpr int XP_LD = 40;               // store transposer: 32 rows x 32 dims bf16, rows padded to 80 bytes
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCALE = 0.17677669529663687f;      // 1 / sqrt(32)

// element offset of chunk c (8 bf16) of head X of row r inside a staged tensor
__device__ __forceinline__ int d_off(int X, int r, int c) {
    return (r >> 3) * 512 + X * 256 + (r & 7) * 32 + 8 * (c ^ ((r >> 2) & 3));
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// token row of slot s of pass `pass`, or -1
struct PassMap {
    int Sp, lg, S, HW, PPT;
    template <bool FULL = false>
    __device__ __forceinline__ long row(int pass, int s) const {
        const int pl = s >> lg, b = s & (Sp - 1), p = pass * PPT + pl;
        return (FULL || (b < S && p < HW)) ? (long)b * HW + p : -1;
    }
};

// Transfers of one pass: NT tensors (q, k, v [, dO]) x 64 rows x 128 bytes.  Wave wv moves the 8-row groups wv and
// 4 + wv of every tensor; lane -> (head lane >> 5, row (lane >> 2) & 7 of the group, chunk position lane & 3).
// Absent rows (image index >= S, pixel >= HW) fetch row 0: finite values that the masks keep out of every result.
template <int NT, bool FULL>
__device__ __forceinline__ void issue_pass(__bf16* __restrict__ st, const __bf16* __restrict__ qkv,
                                           const __bf16* __restrict__ dctx, const PassMap& pm, int pass, int wv, int lane) {
    const int r7 = (lane >> 2) & 7, cp = lane & 3, Xl = lane >> 5;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int grp = h2 * 4 + wv, r = grp * 8 + r7;
        long n = pm.row<FULL>(pass, r);
        if (!FULL && n < 0) n = 0;
        const int c = cp ^ ((r >> 2) & 3);
        const __bf16* s3 = qkv + n * 192 + 32 * Xl + 8 * c;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const __bf16* src = (t < 3) ? s3 + 64 * t : dctx + n * 64 + 32 * Xl + 8 * c;
            ksr_dma16<true>(src, st + t * TEN + grp * 512);      // nt: every row is read once
        }
    }
}

// acc[row = slot of block][col = lane] += sum over the 32 dims of head X of rows[rbase + slot][dim] * x[dim]
__device__ __forceinline__ void rows_times_x(const __bf16* __restrict__ rows, int X, int rbase, int tok, int g,
                                             const bf16x8 (&x)[2], f32x16& acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = mfma16(*reinterpret_cast<const bf16x8*>(rows + d_off(X, rbase + tok, 2 * s + g)), x[s], acc);
}

// acc[row = dim][col = lane] += sum over the 32 slots of a block of rows[cbase + slot][dim] * y[slot], y an
// accumulator tile (register r of lane half g <-> slot (r&3) + 8(r>>2) + 4g).  Every lane of the wave must be active.
__device__ __forceinline__ void tr_times_acc(const __bf16* __restrict__ rows, int X, int cbase, int lane, const f32x16& y,
                                             f32x16& acc) {
    const int g = lane >> 5, db = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r0 = cbase + 16 * s + 4 * g + q;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(rows + d_off(X, r0, 2 * db + (p >> 1)) + 4 * (p & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(rows + d_off(X, r0 + 8, 2 * db + (p >> 1)) + 4 * (p & 1)));
        union { s16x4 h[2]; bf16x8 v; } a;
        a.h[0] = lo; a.h[1] = hi;
        bf16x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)y[8 * s + j];
        acc = mfma16(a.v, b, acc);
    }
}

// this lane's own row as MFMA B operand: x[s][j] = dim 16 s + 8 g + j of head X of row `slot`
__device__ __forceinline__ void own_row(const __bf16* __restrict__ rows, int X, int slot, int g, bf16x8 (&x)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) x[s] = *reinterpret_cast<const bf16x8*>(rows + d_off(X, slot, 2 * s + g));
}

// Accumulator tile (rows = the 32 dims of a head, column = slot) * mul -> wave-private bf16 tile [32 slots][32 dims]
__device__ __forceinline__ void tile_to_xp(__bf16* __restrict__ xp, int lane, const f32x16& acc, float mul) {
    const int tok = lane & 31, g = lane >> 5;
    ksr_wave_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (__bf16)(acc[4 * q + i] * mul);
        *reinterpret_cast<bf16x4*>(xp + tok * XP_LD + 8 * q + 4 * g) = w;
    }
    ksr_wave_fence();
}
// ... -> 64-byte pieces of rows n[0] (slot lane >> 2) and n[1] (slot 16 + (lane >> 2)) of a tensor with ld elements per
// row, at channel c0
template <bool FULL = false, bool NT = false>
__device__ __forceinline__ void xp_to_rows(const __bf16* __restrict__ xp, __bf16* __restrict__ base, const long (&n)[2],
                                           int ld, int c0, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + (lane >> 2), c8 = (lane & 3) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(xp + row * XP_LD + c8);
        if (FULL || n[i] >= 0) {
            if constexpr (NT) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(base + n[i] * ld + c0 + c8));
            } else {
                *reinterpret_cast<uint4*>(base + n[i] * ld + c0 + c8) = v;
            }
        }
    }
    ksr_wave_fence();
}

// ------------------------------------------------------------------------------------------ forward
// FULL: every slot of every pass is a token (S == Sp >= 32, HW a multiple of the pixels per pass): no masks
template <int NW, bool DROP, bool FULL>
__global__ void __launch_bounds__(256, 2) attn_fwd_dma_kernel(AttnArgs a) {
    KSR_MIX_SEED(a);
    __shared__ __attribute__((aligned(16))) __bf16 stA[3 * TEN];
    __shared__ __attribute__((aligned(16))) __bf16 stB[3 * TEN];
    __shared__ __attribute__((aligned(16))) __bf16 xps[4 * 32 * XP_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, g = lane >> 5;
    const int X = wv & 1, sb = wv >> 1;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const PassMap pm{Sp, lg, S, HW, TS / Sp};
    const int npass = (HW + pm.PPT - 1) / pm.PPT;
    const int slot = 32 * sb + tok;
    const int pl = slot >> lg, b = slot & (Sp - 1);
    const int kbase0 = (Sp >= 32) ? pl * Sp : sb * 32;
    const SlotMap smap{Sp, lg, Sp >= 32};
    constexpr bool drop = DROP;
    const float keep = 1.f - a.p_drop, inv_keep = drop ? 1.f / keep : 1.f;
    const unsigned thr = ksr_attn_keep_threshold(keep);
    const float cexp = SCALE * LOG2E;
    __bf16* xp = xps + wv * 32 * XP_LD;
    const __bf16* qkv = reinterpret_cast<const __bf16*>(a.qkv);
    __bf16* ctx = reinterpret_cast<__bf16*>(a.ctx);
    const int G = gridDim.x;

    // The result of a pass waits in the wave's transposer tile and is stored after the NEXT pass's barrier: the wait for
    // the transfers at that barrier then finds no store of this wave in flight.
    long pend[2] = {-1, -1};
    auto body = [&](const __bf16* __restrict__ cur, __bf16* __restrict__ nxt, int pass, bool first) {
        ksr_dma_barrier();                     // cur has landed (every wave waited for its own transfers); nxt is free
        if (pass + G < npass) issue_pass<3, FULL>(nxt, qkv, nullptr, pm, pass + G, wv, lane);
        if (!first) xp_to_rows<FULL>(xp, ctx, pend, 64, 32 * X, lane);
        const int p = pass * pm.PPT + pl;
        const bool live = FULL || ((b < S) && (p < HW));
        const long n = live ? ((long)b * HW + p) : 0;
        pend[0] = pm.row<FULL>(pass, 32 * sb + (lane >> 2));
        pend[1] = pm.row<FULL>(pass, 32 * sb + 16 + (lane >> 2));
        const __bf16 *Q = cur, *K = cur + TEN, *V = cur + 2 * TEN;
        bf16x8 qx[2];
        own_row(Q, X, slot, g, qx);
        // ---- scores^T = K q^T, masked softmax (rows split over the lane pair) ----
        f32x16 sc[NW];
        float m = -INFINITY;
        const unsigned dbase = drop ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) {
            sc[kb] = zero16();
            rows_times_x(K, X, kbase0 + 32 * kb, tok, g, qx, sc[kb]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int same, kbb;
                smap.decode(kbase0, kb, key_of_reg(r, g), pl, same, kbb);
                sc[kb][r] = (FULL || (same && (kbb < S))) ? sc[kb][r] : -INFINITY;
                m = fmaxf(m, sc[kb][r]);
            }
        }
        m = ksr_pair_max(m);
        float l = 0.f;
        const float m2 = m * cexp;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // registers r, r + 1 are neighbouring keys (even, odd image index): one hash decides both
                const float e0 = __builtin_amdgcn_exp2f(fmaf(sc[kb][r], cexp, -m2));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(sc[kb][r + 1], cexp, -m2));
                l += e0 + e1;
                sc[kb][r] = e0; sc[kb][r + 1] = e1;
                if (drop) {
                    int same, kbb;
                    smap.decode(kbase0, kb, key_of_reg(r, g), pl, same, kbb);
                    const unsigned x = ksr_attn_keep_word(dbase, kbb);
                    sc[kb][r] = (x & 0xffffu) < thr ? e0 : 0.f;
                    sc[kb][r + 1] = (x >> 16) < thr ? e1 : 0.f;
                }
            }
        const float inv_l = inv_keep / ksr_pair_sum(l);         // 1 / keep of the kept probabilities folded in
        // ---- ctx^T = V^T P^T ----
        f32x16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) tr_times_acc(V, X, kbase0 + 32 * kb, lane, sc[kb], acc);
        tile_to_xp(xp, lane, acc, inv_l);
    };

    int pass = blockIdx.x;
    if (pass < npass) issue_pass<3, FULL>(stA, qkv, nullptr, pm, pass, wv, lane);
    for (; pass < npass; pass += 2 * G) {
        body(stA, stB, pass, pass == (int)blockIdx.x);
        if (pass + G >= npass) break;
        body(stB, stA, pass + G, false);
    }
    ksr_wave_fence();
    if ((int)blockIdx.x < npass) xp_to_rows<FULL>(xp, ctx, pend, 64, 32 * X, lane);
}

// ------------------------------------------------------------------------------------------ in-projection + forward
// The layer's in-projection inside the attention forward: a pass stages the 64 tokens' INPUT rows (8 KB instead of the
// 24 KB of their q, k, v rows), every wave forms q, k, v of its head for its 32 tokens on the matrix cores (12 products,
// bias in the accumulator's initial value), rounds them to bf16 into the q / k / v tile that the attention part reads,
// and the tile is stored to qkv [N,192] (the backward's saved tensor) from LDS.  qkv is written once and never read.
struct QkvAttnArgs {
    const __bf16* x;      // [N,64]
    const float* w;       // [192,64] in_proj_weight
    const float* bias;    // [192]
    __bf16* qkv;          // [N,192] or null
    __bf16* ctx;          // [N,64]
    int S, Sp, HW;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_dev;
};

// element offset of chunk c (8 bf16, c < 8) of row r of a staged input tile: rows of 128 bytes, chunk position
// c ^ ((r >> 1) & 7) (16 consecutive rows reading the same chunk touch 16 different bank quads)
__device__ __forceinline__ int x_off(int r, int c) { return r * 64 + 8 * (c ^ ((r >> 1) & 7)); }

template <bool FULL>
__device__ __forceinline__ void issue_x(__bf16* __restrict__ st, const __bf16* __restrict__ x, const PassMap& pm, int pass,
                                        int wv, int lane) {
    const int r8 = lane >> 3, cp = lane & 7;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int grp = h2 * 4 + wv, r = grp * 8 + r8;
        long n = pm.row<FULL>(pass, r);
        if (!FULL && n < 0) n = 0;
        const int c = cp ^ ((r >> 1) & 7);
        ksr_dma16<false>(x + n * 64 + 8 * c, st + grp * 512);
    }
}

template <int NW, bool DROP, bool FULL>
__global__ void __launch_bounds__(256, 2) qkv_attn_fwd_dma_kernel(QkvAttnArgs a) {
    KSR_MIX_SEED(a);
    __shared__ __attribute__((aligned(16))) __bf16 xsA[TEN];
    __shared__ __attribute__((aligned(16))) __bf16 xsB[TEN];
    __shared__ __attribute__((aligned(16))) __bf16 tile[3 * TEN];           // q, k, v of the pass (d_off layout)
    __shared__ __attribute__((aligned(16))) float bfrag[6 * 2 * 16];        // [out block][lane half][register]
    __shared__ __attribute__((aligned(16))) __bf16 xps[4 * 32 * XP_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, g = lane >> 5;
    const int X = wv & 1, sb = wv >> 1;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const PassMap pm{Sp, lg, S, HW, TS / Sp};
    const int npass = (HW + pm.PPT - 1) / pm.PPT;
    const int slot = 32 * sb + tok;
    const int pl = slot >> lg, b = slot & (Sp - 1);
    const int kbase0 = (Sp >= 32) ? pl * Sp : sb * 32;
    const SlotMap smap{Sp, lg, Sp >= 32};
    const float keep = 1.f - a.p_drop, inv_keep = DROP ? 1.f / keep : 1.f;
    const unsigned thr = ksr_attn_keep_threshold(keep);
    const float cexp = SCALE * LOG2E;
    __bf16* xp = xps + wv * 32 * XP_LD;
    const int G = gridDim.x;

    // this wave's weight fragments stay in registers (48): element i of step s of tensor t =
    // W[64 t + 32 X + (lane & 31)][16 s + 8 (lane >> 5) + i]
    bf16x8 wreg[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float* wp = a.w + (64 * t + 32 * X + tok) * 64 + 16 * s4 + 8 * g;
            const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
            const float wf[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) wreg[t][s4][i] = (__bf16)wf[i];
        }
    for (int idx = tid; idx < 6 * 2 * 16; idx += 256)
        bfrag[idx] = a.bias[32 * (idx >> 5) + key_of_reg(idx & 15, (idx >> 4) & 1)];

    long pend[2] = {-1, -1};
    // since the transfers of this pass were issued (top of the previous pass) the wave has issued 2 stores of the context
    // (not in the pass after the first) and 6 of qkv (if it is written): those may stay in flight
    const bool save_qkv = a.qkv != nullptr;
    auto body = [&](const __bf16* __restrict__ cur, __bf16* __restrict__ nxt, int pass, bool first, bool second) {
        // cur has landed; every wave is done with the previous pass's tile
        if (FULL && !first && !second) {
            if (save_qkv) ksr_dma_barrier_but<8>(); else ksr_dma_barrier_but<2>();
        } else {
            ksr_dma_barrier();
        }
        if (pass + G < npass) issue_x<FULL>(nxt, a.x, pm, pass + G, wv, lane);
        if (!first) xp_to_rows<FULL>(xp, a.ctx, pend, 64, 32 * X, lane);
        const int p = pass * pm.PPT + pl;
        const bool live = FULL || ((b < S) && (p < HW));
        const long n = live ? ((long)b * HW + p) : 0;
        pend[0] = pm.row<FULL>(pass, 32 * sb + (lane >> 2));
        pend[1] = pm.row<FULL>(pass, 32 * sb + 16 + (lane >> 2));
        // ---- q, k, v of head X for this wave's 32 tokens ----
        {
            bf16x8 xb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) xb[s] = *reinterpret_cast<const bf16x8*>(cur + x_off(slot, 2 * s + g));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int ob = 2 * t + X;
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&bfrag[(ob * 2 + g) * 16 + 4 * q]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[4 * q + i] = bv[i];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = mfma16(wreg[t][s], xb[s], acc);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = (__bf16)acc[4 * q + i];
                    *reinterpret_cast<bf16x4*>(tile + t * TEN + d_off(X, slot, q) + 4 * g) = w;
                }
            }
        }
        ksr_lds_barrier();                     // the pass's q, k, v tile is complete
        if (save_qkv) {
            // tile -> qkv rows: piece f of the tile in LDS order = (tensor, 8-row group, head, row & 7, chunk position)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int f = i * 256 + tid;
                const int t = f >> 9, r = ((f >> 6) & 7) * 8 + ((f >> 2) & 7), Xf = (f >> 5) & 1, c = (f & 3) ^ ((r >> 2) & 3);
                const uint4 v = *reinterpret_cast<const uint4*>(tile + 8 * f);
                const long nr = pm.row<FULL>(pass, r);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                if (FULL || nr >= 0)      // non-temporal: read next by the backward, a whole forward and loss later
                    __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(a.qkv + nr * 192 + 64 * t + 32 * Xf + 8 * c));
            }
        }
        const __bf16 *Q = tile, *K = tile + TEN, *V = tile + 2 * TEN;
        bf16x8 qx[2];
        own_row(Q, X, slot, g, qx);
        f32x16 sc[NW];
        float m = -INFINITY;
        const unsigned dbase = DROP ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) {
            sc[kb] = zero16();
            rows_times_x(K, X, kbase0 + 32 * kb, tok, g, qx, sc[kb]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int same, kbb;
                smap.decode(kbase0, kb, key_of_reg(r, g), pl, same, kbb);
                sc[kb][r] = (FULL || (same && (kbb < S))) ? sc[kb][r] : -INFINITY;
                m = fmaxf(m, sc[kb][r]);
            }
        }
        m = ksr_pair_max(m);
        float l = 0.f;
        const float m2 = m * cexp;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(sc[kb][r], cexp, -m2));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(sc[kb][r + 1], cexp, -m2));
                l += e0 + e1;
                sc[kb][r] = e0; sc[kb][r + 1] = e1;
                if (DROP) {
                    int same, kbb;
                    smap.decode(kbase0, kb, key_of_reg(r, g), pl, same, kbb);
                    const unsigned x = ksr_attn_keep_word(dbase, kbb);
                    sc[kb][r] = (x & 0xffffu) < thr ? e0 : 0.f;
                    sc[kb][r + 1] = (x >> 16) < thr ? e1 : 0.f;
                }
            }
        const float inv_l = inv_keep / ksr_pair_sum(l);
        f32x16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) tr_times_acc(V, X, kbase0 + 32 * kb, lane, sc[kb], acc);
        tile_to_xp(xp, lane, acc, inv_l);
    };

    int pass = blockIdx.x;
    if (pass < npass) issue_x<FULL>(xsA, a.x, pm, pass, wv, lane);
    for (; pass < npass; pass += 2 * G) {
        body(xsA, xsB, pass, pass == (int)blockIdx.x, false);
        if (pass + G >= npass) break;
        body(xsB, xsA, pass + G, false, pass == (int)blockIdx.x);
    }
    ksr_wave_fence();
    if ((int)blockIdx.x < npass) xp_to_rows<FULL>(xp, a.ctx, pend, 64, 32 * X, lane);
}

// ------------------------------------------------------------------------------------------ backward
template <int NW, bool DROP, bool FULL>
__global__ void __launch_bounds__(256, 2) attn_bwd_dma_kernel(AttnBwdArgs a) {
    KSR_MIX_SEED(a);
    __shared__ __attribute__((aligned(16))) __bf16 stA[4 * TEN];
    __shared__ __attribute__((aligned(16))) __bf16 stB[4 * TEN];
    __shared__ __attribute__((aligned(16))) float stat[2 * TS * 4];   // per head and query slot: m' + log2(l keep), Dk, keep bits of the lane halves
    __shared__ __attribute__((aligned(16))) __bf16 xps[4 * 32 * XP_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, g = lane >> 5;
    const int X = wv & 1, sb = wv >> 1;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const PassMap pm{Sp, lg, S, HW, TS / Sp};
    const int npass = (HW + pm.PPT - 1) / pm.PPT;
    const int slot = 32 * sb + tok;
    const int pl = slot >> lg, b = slot & (Sp - 1);
    const int base0 = (Sp >= 32) ? pl * Sp : sb * 32;
    const SlotMap smap{Sp, lg, Sp >= 32};
    const float keep = 1.f - a.p_drop, inv_keep = DROP ? 1.f / keep : 1.f;
    const unsigned thr = ksr_attn_keep_threshold(keep);
    // a query lane (half gq) holds key tok of its block kb in register (tok & 3) + 4 (tok >> 3), gq = (tok >> 2) & 1
    const bool my_half = (tok >> 2) & 1;
    const unsigned my_bit = 1u << ((tok & 3) + 4 * (tok >> 3) + (NW == 2 ? 16 * sb : 0));
    const float cexp = SCALE * LOG2E;
    __bf16* xp = xps + wv * 32 * XP_LD;
    float* stx = stat + X * TS * 4;
    const __bf16* qkv = reinterpret_cast<const __bf16*>(a.qkv);
    const __bf16* dctx = reinterpret_cast<const __bf16*>(a.dctx);
    __bf16* dqkv = reinterpret_cast<__bf16*>(a.dqkv);
    const int G = gridDim.x;

    auto body = [&](const __bf16* __restrict__ cur, __bf16* __restrict__ nxt, int pass, bool first) {
        // cur has landed (its transfers were issued at the top of the previous pass; the 6 stores of that pass's results
        // are younger and may stay in flight); every wave is done with the previous pass: nxt (its stage) is free
        if (FULL && !first) ksr_dma_barrier_but<6>(); else ksr_dma_barrier();
        if (pass + G < npass) issue_pass<4, FULL>(nxt, qkv, dctx, pm, pass + G, wv, lane);
        const int p = pass * pm.PPT + pl;
        const bool live = FULL || ((b < S) && (p < HW));
        const long n = live ? ((long)b * HW + p) : 0;
        long nrow[2];
        nrow[0] = pm.row<FULL>(pass, 32 * sb + (lane >> 2));
        nrow[1] = pm.row<FULL>(pass, 32 * sb + 16 + (lane >> 2));
        const __bf16 *Q = cur, *K = cur + TEN, *V = cur + 2 * TEN, *DO = cur + 3 * TEN;
        // ---------------- phase A: this lane is a query ----------------
        {
            bf16x8 qx[2], dox[2];
            own_row(Q, X, slot, g, qx);
            own_row(DO, X, slot, g, dox);
            float m = -INFINITY;
            const unsigned dbase = DROP ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
            f32x16 st[NW], dp[NW];
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
                st[kb] = zero16();
                rows_times_x(K, X, base0 + 32 * kb, tok, g, qx, st[kb]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int same, kbb;
                    smap.decode(base0, kb, key_of_reg(r, g), pl, same, kbb);
                    st[kb][r] = (FULL || (same && (kbb < S))) ? st[kb][r] : -INFINITY;
                    m = fmaxf(m, st[kb][r]);
                }
            }
            m = ksr_pair_max(m);
            float l = 0.f, dsum = 0.f;
            const float m2 = m * cexp;
            unsigned kw = 0u;                  // keep bits of this lane's keys: bit r + 16 kb
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
                dp[kb] = zero16();
                rows_times_x(V, X, base0 + 32 * kb, tok, g, dox, dp[kb]);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // registers r, r + 1 are neighbouring keys (even, odd image index): one hash decides both
                    const float e0 = __builtin_amdgcn_exp2f(fmaf(st[kb][r], cexp, -m2));
                    const float e1 = __builtin_amdgcn_exp2f(fmaf(st[kb][r + 1], cexp, -m2));
                    st[kb][r] = e0; st[kb][r + 1] = e1;
                    l += e0 + e1;
                    if (DROP) {
                        int same, kbb;
                        smap.decode(base0, kb, key_of_reg(r, g), pl, same, kbb);
                        const unsigned x = ksr_attn_keep_word(dbase, kbb);
                        const bool k0 = (x & 0xffffu) < thr, k1 = (x >> 16) < thr;
                        dp[kb][r] = k0 ? dp[kb][r] : 0.f;
                        dp[kb][r + 1] = k1 ? dp[kb][r + 1] : 0.f;
                        kw |= (k0 ? 1u << (r + 16 * kb) : 0u) | (k1 ? 2u << (r + 16 * kb) : 0u);
                    }
                    dsum = fmaf(e0, dp[kb][r], dsum);
                    dsum = fmaf(e1, dp[kb][r + 1], dsum);
                }
            }
            // With dP' = dP on the kept keys, 0 elsewhere (no 1/keep): D = rowsum(dO * O) = Dk / keep with
            // Dk = sum_j P_ij dP'_ij, and dS_ij = P_ij (dP'_ij / keep - D) = (P_ij / keep) (dP'_ij - Dk).
            const float inv_l = 1.f / ksr_pair_sum(l);
            const float Dk = ksr_pair_sum(dsum) * inv_l;
            const float c2 = inv_l * SCALE * inv_keep, dc2 = -Dk * c2;
            f32x16 dq = zero16();
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[kb][r] = st[kb][r] * fmaf(dp[kb][r], c2, dc2);      // dS (scaled for dq)
                tr_times_acc(K, X, base0 + 32 * kb, lane, dp[kb], dq);
            }
            // the key side recomputes P_ij / keep = exp2(s c - (m c + log2 l + log2 keep)) and reads Dk and its keep bit
            if (g == 0) *reinterpret_cast<f32x2*>(&stx[slot * 4]) = f32x2{m2 - __builtin_amdgcn_logf(inv_l * inv_keep), Dk};
            if (DROP) stx[slot * 4 + 2 + g] = __uint_as_float(kw);
            tile_to_xp(xp, lane, dq, 1.f);
        }
        __builtin_amdgcn_sched_barrier(0);
        ksr_lds_barrier();                     // the statistics are visible
        xp_to_rows<FULL>(xp, dqkv, nrow, 192, 32 * X, lane);
        // ---------------- phase B: this lane is a key ----------------
        {
            bf16x8 kx[2], vx[2];
            own_row(K, X, slot, g, kx);
            own_row(V, X, slot, g, vx);
            f32x16 dk = zero16(), dv = zero16();
#pragma unroll
            for (int qblk = 0; qblk < NW; ++qblk) {
                f32x16 sq = zero16(), dp = zero16();
                rows_times_x(Q, X, base0 + 32 * qblk, tok, g, kx, sq);
                rows_times_x(DO, X, base0 + 32 * qblk, tok, g, vx, dp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qs = base0 + 32 * qblk + key_of_reg(r, g);      // query slot
                    int same, qbb;
                    smap.decode(base0, qblk, key_of_reg(r, g), pl, same, qbb);
                    const bool ok = FULL || (same && (qbb < S) && (p < HW));
                    const f32x4 stt = *reinterpret_cast<const f32x4*>(&stx[qs * 4]);
                    const float pij = ok ? __builtin_amdgcn_exp2f(fmaf(sq[r], cexp, -stt[0])) : 0.f;      // P_ij / keep
                    float pt = pij, dpr = dp[r];
                    if (DROP) {
                        // bit of (query qs, this key) in the word of the query lane half that holds this key
                        const bool kept = (__float_as_uint(my_half ? stt[3] : stt[2]) & my_bit) != 0u;
                        pt = kept ? pij : 0.f;
                        dpr = kept ? dpr : 0.f;
                    }
                    sq[r] = pij * (dpr - stt[1]);    // dS
                    dp[r] = pt;                       // probabilities after dropout
                }
                tr_times_acc(Q, X, base0 + 32 * qblk, lane, sq, dk);
                tr_times_acc(DO, X, base0 + 32 * qblk, lane, dp, dv);
            }
            tile_to_xp(xp, lane, dk, SCALE);
            xp_to_rows<FULL>(xp, dqkv, nrow, 192, 64 + 32 * X, lane);
            tile_to_xp(xp, lane, dv, 1.f);
            xp_to_rows<FULL>(xp, dqkv, nrow, 192, 128 + 32 * X, lane);
        }
    };

    int pass = blockIdx.x;
    if (pass < npass) issue_pass<4, FULL>(stA, qkv, dctx, pm, pass, wv, lane);
    for (; pass < npass; pass += 2 * G) {
        body(stA, stB, pass, pass == (int)blockIdx.x);
        if (pass + G >= npass) break;
        body(stB, stA, pass + G, false);
    }
}

}  // namespace

bool ksr_attn_dma_supported(int Sp) { return Sp >= 4 && Sp <= 64; }

int ksr_attn_dma_fwd_launch(const AttnArgs& a, hipStream_t stream) {
    const int ppt = TS / a.Sp, npass = (a.HW + ppt - 1) / ppt;
    const dim3 g((unsigned)(npass < 512 ? npass : 512)), block(256);
    const bool full = a.S == a.Sp && a.Sp >= 32 && a.HW % ppt == 0;
    const bool drop = a.p_drop > 0.f;
#define KSR_ATTN_FWD(NWV, DR)                                                                                \
    do {                                                                                                     \
        if (full) hipLaunchKernelGGL((attn_fwd_dma_kernel<NWV, DR, true>), g, block, 0, stream, a);          \
        else hipLaunchKernelGGL((attn_fwd_dma_kernel<NWV, DR, false>), g, block, 0, stream, a);              \
    } while (0)
    if (a.Sp == 64) {
        if (drop) KSR_ATTN_FWD(2, true); else KSR_ATTN_FWD(2, false);
    } else {
        if (drop) KSR_ATTN_FWD(1, true); else KSR_ATTN_FWD(1, false);
    }
#undef KSR_ATTN_FWD
    KSR_LAUNCH_CHECK("batch_attn_fwd(bf16 storage)");
    return KSR_OK;
}

int ksr_qkv_attn_dma_fwd_launch(const float* x, const float* w, const float* bias, float* qkv, float* ctx, int S, int Sp,
                                int HW, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                                hipStream_t stream) {
    QkvAttnArgs a{reinterpret_cast<const __bf16*>(x), w, bias, reinterpret_cast<__bf16*>(qkv), reinterpret_cast<__bf16*>(ctx),
                  S, Sp, HW, p_drop, seed, seed_dev};
    const int ppt = TS / Sp, npass = (HW + ppt - 1) / ppt;
    const dim3 g((unsigned)(npass < 512 ? npass : 512)), block(256);
    const bool full = S == Sp && Sp >= 32 && HW % ppt == 0;
    const bool drop = p_drop > 0.f;
#define KSR_QKV_ATTN(NWV, DR)                                                                                    \
    do {                                                                                                         \
        if (full) hipLaunchKernelGGL((qkv_attn_fwd_dma_kernel<NWV, DR, true>), g, block, 0, stream, a);          \
        else hipLaunchKernelGGL((qkv_attn_fwd_dma_kernel<NWV, DR, false>), g, block, 0, stream, a);              \
    } while (0)
    if (Sp == 64) {
        if (drop) KSR_QKV_ATTN(2, true); else KSR_QKV_ATTN(2, false);
    } else {
        if (drop) KSR_QKV_ATTN(1, true); else KSR_QKV_ATTN(1, false);
    }
#undef KSR_QKV_ATTN
    KSR_LAUNCH_CHECK("batch_qkv_attn_fwd(bf16 storage)");
    return KSR_OK;
}

int ksr_attn_dma_bwd_launch(const AttnBwdArgs& a, hipStream_t stream) {
    const int ppt = TS / a.Sp, npass = (a.HW + ppt - 1) / ppt;
    const dim3 g((unsigned)(npass < 512 ? npass : 512)), block(256);
    const bool drop = a.p_drop > 0.f;
    const bool full = a.S == a.Sp && a.Sp >= 32 && a.HW % ppt == 0;
#define KSR_ATTN_BWD(NWV, DR)                                                                                \
    do {                                                                                                     \
        if (full) hipLaunchKernelGGL((attn_bwd_dma_kernel<NWV, DR, true>), g, block, 0, stream, a);          \
        else hipLaunchKernelGGL((attn_bwd_dma_kernel<NWV, DR, false>), g, block, 0, stream, a);              \
    } while (0)
    if (a.Sp == 64) {
        if (drop) KSR_ATTN_BWD(2, true); else KSR_ATTN_BWD(2, false);
    } else {
        if (drop) KSR_ATTN_BWD(1, true); else KSR_ATTN_BWD(1, false);
    }
#undef KSR_ATTN_BWD
    KSR_LAUNCH_CHECK("batch_attn_bwd(bf16 storage)");
    return KSR_OK;
}
