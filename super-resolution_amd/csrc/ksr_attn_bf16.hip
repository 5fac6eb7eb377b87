// Batch-axis attention with bf16 MFMA operands (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the bf16-mode
// variants of the kernels in ksr_attn.hip.  Same slot scheme (slot = pl*Sp + b, a wave owns 32 slots, the
// query sits on the MFMA column axis, the accumulator tile of one product is the B operand of the next) and
// the same fp32 softmax / dropout arithmetic; only the four (forward) / ten (backward) products per key block
// are rounded to bf16.  A 16-deep MFMA step pairs k-slot j of lane half g with
//   dims   16*s + 8*g + j                          for products over the 32 dims of a head, and
//   slots  16*s + 8*(j>>2) + 4*g + (j&3)           for products over the 32 slots of a block
// (the second is the order in which an accumulator tile holds its rows).
// LDS holds every operand once, as rows of the 32 dims of one head (64 bytes, no padding): a product over dims
// reads a row with one 16-byte access per step; a product over slots reads the same rows with the transposing
// load ds_read_b64_tr_b16 (each 16-lane group fetches a 4-slot x 16-dim block and lane i receives dim i of the 4
// slots), two per step.  The four 16-byte chunks of row r are stored at chunk position c ^ ((r >> 2) & 3), which
// makes the row writes, the row reads and the transposing reads all bank-conflict free.
// Rows of qkv / ctx are fetched with lane-linear 16-byte accesses (16 lanes per 256-byte row) and
// redistributed through a wave-private LDS tile.
#include "ksr_attn.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// element offset of chunk c (8 bf16) of row r in a swizzled row array
__device__ __forceinline__ int sw_off(int r, int c) { return r * 32 + 8 * (c ^ ((r >> 2) & 3)); }

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr float LOG2E = 1.4426950408889634f;
// exp(x - m) as one FMA and one v_exp_f32: m2 = m * log2(e) is formed once per row
__device__ __forceinline__ float fast_exp_m(float x, float m2) { return __builtin_amdgcn_exp2f(fmaf(x, LOG2E, -m2)); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// Row (token) of every slot of this wave, or -1: slot = 32*wave + row.
struct RowMap {
    int Sp, lg, big, S, HW, wv, pl_w, b0_w, tile, PPW;
    __device__ __forceinline__ long operator()(int row) const {
        int pl, b;
        if (big) { pl = pl_w; b = b0_w + row; }
        else { const int s = wv * 32 + row; pl = s >> lg; b = s & (Sp - 1); }
        const int p = tile * PPW + pl;
        return (b < S && p < HW) ? (long)b * HW + p : -1;
    }
};

// Lane-linear fetch of 32 rows x 64 channels (channels c0.. of a tensor with ld elements per row, stored as T).
// SELECT: no branch (absent rows read row 0 and are zeroed afterwards) - fewer stalls where the registers allow it.
template <class T, bool SELECT = false>
__device__ __forceinline__ void fetch_rows(const void* __restrict__ base, const RowMap& rm, int ld, int c0, int lane,
                                           KsrTile<T>& t) {
    if constexpr (SELECT) {
        if constexpr (sizeof(T) == 4) {
            const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long n = rm(4 * i + r4);
                const bool ok = n >= 0;
                const float4 v = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + (ok ? n : 0) * ld + c0 + c4));
                t.v[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        } else {
            const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long n = rm(8 * i + r8);
                const bool ok = n >= 0;
                const uint4 v = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + (ok ? n : 0) * ld + c0 + c8));
                t.v[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long n = rm(4 * i + r4);
            t.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n >= 0) t.v[i] = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + n * ld + c0 + c4));
        }
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long n = rm(8 * i + r8);
            t.v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (n >= 0) t.v[i] = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + n * ld + c0 + c8));
        }
    }
}

// Fetched tile -> this lane's operand form: x[X][s][j] = channel 32*X + 16*s + 8*g + j of row (lane & 31).
template <class T>
__device__ __forceinline__ void finish_split(float* __restrict__ xb, int lane, const KsrTile<T>& t, float (&x)[2][2][8]) {
    ksr_wave_fence();
    ksr_co_to_lds<T>(xb, lane, t);
    ksr_wave_fence();
    const float* rp = xb + (lane & 31) * KSR_XPOSE_LD + 8 * (lane >> 5);
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 t0 = *reinterpret_cast<const float4*>(rp + 32 * X + 16 * s);
            const float4 t1 = *reinterpret_cast<const float4*>(rp + 32 * X + 16 * s + 4);
            x[X][s][0] = t0.x; x[X][s][1] = t0.y; x[X][s][2] = t0.z; x[X][s][3] = t0.w;
            x[X][s][4] = t1.x; x[X][s][5] = t1.y; x[X][s][6] = t1.z; x[X][s][7] = t1.w;
        }
}

__device__ __forceinline__ void pack_split(const float (&x)[2][2][8], float mul, bf16x8 (&p)[2][2]) {
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) p[X][s][j] = (__bf16)(x[X][s][j] * mul);
}

// acc[row = slot of block][col = lane] += sum over the 32 dims of rows[(rbase + slot)][dim] * x[dim]
__device__ __forceinline__ void rows_times_x(const __bf16* __restrict__ rows, int rbase, int tok, int g,
                                             const bf16x8 (&x)[2], f32x16& acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = mfma16(*reinterpret_cast<const bf16x8*>(rows + sw_off(rbase + tok, 2 * s + g)), x[s], acc);
}

// acc[row = dim][col = lane] += sum over the 32 slots of a block of rows[cbase + slot][dim] * y[slot], y an
// accumulator tile (register r of lane half g <-> slot (r&3) + 8(r>>2) + 4g).  Every lane of the wave must be active.
__device__ __forceinline__ void tr_times_acc(const __bf16* __restrict__ rows, int cbase, int lane, const f32x16& y,
                                             f32x16& acc) {
    const int g = lane >> 5, db = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r0 = cbase + 16 * s + 4 * g + q;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(rows + sw_off(r0, 2 * db + (p >> 1)) + 4 * (p & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(rows + sw_off(r0 + 8, 2 * db + (p >> 1)) + 4 * (p & 1)));
        union { s16x4 h[2]; bf16x8 v; } a;
        a.h[0] = lo; a.h[1] = hi;
        bf16x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)y[8 * s + j];
        acc = mfma16(a.v, b, acc);
    }
}

// Accumulator tile (rows = 32 dims of one head) -> 32-float-wide LDS tile -> rows of a tensor, channels c0..c0+31
template <class T>
__device__ __forceinline__ void store_dims32(float* __restrict__ xb, void* __restrict__ base, const RowMap& rm, int ld,
                                             int c0, int lane, const f32x16& acc, float mul) {
    const int tok = lane & 31, g = lane >> 5;
    ksr_wave_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(xb + tok * KSR_XPOSE16_LD + 8 * q + 4 * g) =
            make_float4(acc[4 * q] * mul, acc[4 * q + 1] * mul, acc[4 * q + 2] * mul, acc[4 * q + 3] * mul);
    ksr_wave_fence();
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            const float4 t = *reinterpret_cast<const float4*>(xb + row * KSR_XPOSE16_LD + c4);
            const long n = rm(row);
            if (n >= 0) *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + n * ld + c0 + c4) = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane, row = idx >> 2, c8 = (idx & 3) * 8;
            const float* sp = xb + row * KSR_XPOSE16_LD + c8;
            const float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const long n = rm(row);
            if (n >= 0) *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(base) + n * ld + c0 + c8) = ksr_pack8(x);
        }
    }
}

// ------------------------------------------------------------------------------------------ forward
struct __attribute__((aligned(16))) FwdSmem {
    __bf16 k[2][128 * 32];     // K rows per head (swizzled chunks, see sw_off)
    __bf16 v[2][128 * 32];     // V rows per head
    float xpose[4 * KSR_XPOSE_FLOATS];
};

template <int NW, class ST>
__global__ void __launch_bounds__(256, 2) attn_fwd_bf16_kernel(AttnArgs a) {
    KSR_MIX_SEED(a);
    __shared__ FwdSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, g = lane >> 5;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    const int PPW = 128 / Sp;
    const int ntiles = (HW + PPW - 1) / PPW;
    const int slot = wv * 32 + tok;
    const int pl = slot / Sp, b = slot - pl * Sp;
    const float scale = 0.17677669529663687f;
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const bool drop = a.p_drop > 0.f;
    const int kbase0 = (Sp >= 32) ? pl * Sp : wv * 32;
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const SlotMap smap{Sp, lg, NW > 1 ? 1 : (Sp >= 32)};     // more than one block per query implies Sp >= 64
    float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const RowMap rm{Sp, lg, NW > 1 ? true : (Sp >= 32), S, HW, wv, (wv * 32) / Sp, (wv * 32) % Sp, tile, PPW};
        const int p = tile * PPW + pl;
        const bool live = (b < S) && (p < HW);
        const long n = live ? ((long)b * HW + p) : 0;

        bf16x8 qb[2][2];
        {
            KsrTile<ST> fq, fk, fv;
            fetch_rows<ST, true>(a.qkv, rm, 192, 0, lane, fq);
            fetch_rows<ST, true>(a.qkv, rm, 192, 64, lane, fk);
            fetch_rows<ST, true>(a.qkv, rm, 192, 128, lane, fv);
            float x[2][2][8];
            finish_split<ST>(xb, lane, fq, x);
            pack_split(x, scale, qb);
            __syncthreads();                        // previous tile's readers of k / v are done
            finish_split<ST>(xb, lane, fk, x);
#pragma unroll
            for (int X = 0; X < 2; ++X)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 w;
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (__bf16)x[X][s][j];
                    *reinterpret_cast<bf16x8*>(&sm.k[X][sw_off(slot, 2 * s + g)]) = w;
                }
            finish_split<ST>(xb, lane, fv, x);
#pragma unroll
            for (int X = 0; X < 2; ++X)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 w;
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (__bf16)x[X][s][j];
                    *reinterpret_cast<bf16x8*>(&sm.v[X][sw_off(slot, 2 * s + g)]) = w;
                }
        }
        __syncthreads();

#pragma unroll
        for (int X = 0; X < 2; ++X) {
            // ---- scores^T = K q^T, masked softmax (rows split over the lane pair) ----
            f32x16 sc[NW];
            float m = -INFINITY;
            const unsigned dbase = drop ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
                sc[kb] = zero16();
                rows_times_x(sm.k[X], kbase0 + 32 * kb, tok, g, qb[X], sc[kb]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int same, kbb;
                    smap.decode(kbase0, kb, key_of_reg(r, g), pl, same, kbb);
                    sc[kb][r] = (same && (kbb < S)) ? sc[kb][r] : -INFINITY;
                    m = fmaxf(m, sc[kb][r]);
                }
            }
            m = ksr_pair_max(m);
            float l = 0.f;
            const float m2 = m * LOG2E;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pj = fast_exp_m(sc[kb][r], m2);
                    l += pj;
                    if (drop) {
                        int same, kbb;
                        smap.decode(kbase0, kb, key_of_reg(r, g), pl, same, kbb);
                        pj = ksr_attn_keep(dbase, kbb, keep) ? pj * inv_keep : 0.f;
                    }
                    sc[kb][r] = pj;
                }
            const float inv_l = 1.f / ksr_pair_sum(l);
            // ---- ctx^T = V^T P^T ----
            f32x16 acc = zero16();
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) tr_times_acc(sm.v[X], kbase0 + 32 * kb, lane, sc[kb], acc);
            store_dims32<ST>(xb, a.ctx, rm, 64, 32 * X, lane, acc, inv_l);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
struct __attribute__((aligned(16))) BwdSmem {
    __bf16 rowA[128 * 32];    // phase A: K rows  | phase B: Q rows (scaled); swizzled chunks, see sw_off
    __bf16 rowB[128 * 32];    // phase A: V rows  | phase B: dO rows
    float stat[128 * 4];      // per query slot: m log2(e) + log2(l), unused, D, dropout base (bits)
    float xpose[4 * KSR_XPOSE_FLOATS];
};

__device__ __forceinline__ void put_rows(__bf16* __restrict__ rows, int slot, int g, const bf16x8 (&x)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(&rows[sw_off(slot, 2 * s + g)]) = x[s];
}

template <int NW, bool DROP, class ST>
__global__ void __launch_bounds__(256, 2) attn_bwd_bf16_kernel(AttnBwdArgs a) {
    KSR_MIX_SEED(a);
    __shared__ BwdSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, g = lane >> 5;
    const int S = a.S, Sp = a.Sp, HW = a.HW;
    const int PPW = 128 / Sp;
    const int ntiles = (HW + PPW - 1) / PPW;
    const int slot = wv * 32 + tok;
    const int pl = slot / Sp, b = slot - pl * Sp;
    const float scale = 0.17677669529663687f;
    const float keep = 1.f - a.p_drop, inv_keep = 1.f / keep;
    const int base0 = (Sp >= 32) ? pl * Sp : wv * 32;
    int lg = 0;
    while ((1 << lg) < Sp) ++lg;
    const SlotMap smap{Sp, lg, NW > 1 ? 1 : (Sp >= 32)};     // more than one block per query implies Sp >= 64
    float* xb = sm.xpose + wv * KSR_XPOSE_FLOATS;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const RowMap rm{Sp, lg, NW > 1 ? true : (Sp >= 32), S, HW, wv, (wv * 32) / Sp, (wv * 32) % Sp, tile, PPW};
        const int p = tile * PPW + pl;
        const bool live = (b < S) && (p < HW);
        const long n = live ? ((long)b * HW + p) : 0;

        // operands of this lane's token in MFMA form, both heads; D = rowsum(dO * O) per head (fp32)
        bf16x8 qb[2][2], kb_[2][2], vb[2][2], dob[2][2];
        float D[2];
        {
            KsrTile<ST> f0, f1, f2;
            float x[2][2][8];
            fetch_rows<ST>(a.qkv, rm, 192, 0, lane, f0);
            fetch_rows<ST>(a.qkv, rm, 192, 64, lane, f1);
            fetch_rows<ST>(a.qkv, rm, 192, 128, lane, f2);
            finish_split<ST>(xb, lane, f0, x); pack_split(x, scale, qb);
            finish_split<ST>(xb, lane, f1, x); pack_split(x, 1.f, kb_);
            finish_split<ST>(xb, lane, f2, x); pack_split(x, 1.f, vb);
            fetch_rows<ST>(a.dctx, rm, 64, 0, lane, f0);
            fetch_rows<ST>(a.ctx, rm, 64, 0, lane, f1);
            float o[2][2][8];
            finish_split<ST>(xb, lane, f0, x); pack_split(x, 1.f, dob);
            finish_split<ST>(xb, lane, f1, o);
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                float d = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += x[X][s][j] * o[X][s][j];
                D[X] = ksr_pair_sum(d);
            }
        }

#pragma unroll 1
        for (int X = 0; X < 2; ++X) {
            // this head's operands sit in slot 0: head 1 moves down at the end of the first pass, so that the
            // (not unrolled) loop body needs no second copy of them
            bf16x8 (&qx)[2] = qb[0], (&kx)[2] = kb_[0], (&vx)[2] = vb[0], (&dox)[2] = dob[0];
            const float DX = D[0];
            // ---------------- phase A: this lane is a query; LDS holds K rows and V rows ----------------
            __syncthreads();
            put_rows(sm.rowA, slot, g, kx);
            put_rows(sm.rowB, slot, g, vx);
            __syncthreads();
            float m = -INFINITY, inv_l;
            const unsigned dbase = DROP ? ksr_attn_drop_base(a.seed, (unsigned long long)n, X) : 0u;
            {
                f32x16 st[NW];
#pragma unroll
                for (int kb = 0; kb < NW; ++kb) {
                    st[kb] = zero16();
                    rows_times_x(sm.rowA, base0 + 32 * kb, tok, g, qx, st[kb]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int same, kbb;
                        smap.decode(base0, kb, key_of_reg(r, g), pl, same, kbb);
                        st[kb][r] = (same && (kbb < S)) ? st[kb][r] : -INFINITY;
                        m = fmaxf(m, st[kb][r]);
                    }
                }
                m = ksr_pair_max(m);
                float l = 0.f;
                const float m2 = m * LOG2E;
#pragma unroll
                for (int kb = 0; kb < NW; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { st[kb][r] = fast_exp_m(st[kb][r], m2); l += st[kb][r]; }
                inv_l = 1.f / ksr_pair_sum(l);
                const float ils = inv_l * scale;
                f32x16 dq = zero16();
#pragma unroll
                for (int kb = 0; kb < NW; ++kb) {
                    f32x16 dp = zero16();
                    rows_times_x(sm.rowB, base0 + 32 * kb, tok, g, dox, dp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float dpr = dp[r];
                        if (DROP) {
                            int same, kbb;
                            smap.decode(base0, kb, key_of_reg(r, g), pl, same, kbb);
                            dpr = ksr_attn_keep(dbase, kbb, keep) ? dpr * inv_keep : 0.f;
                        }
                        dp[r] = st[kb][r] * ils * (dpr - DX);      // dS (scaled for dq)
                    }
                    tr_times_acc(sm.rowA, base0 + 32 * kb, lane, dp, dq);
                }
                store_dims32<ST>(xb, a.dqkv, rm, 192, 32 * X, lane, dq, 1.f);
            }
            // ---------------- phase B: this lane is a key; LDS holds Q rows, dO rows, stats ----------------
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            put_rows(sm.rowA, slot, g, qx);
            put_rows(sm.rowB, slot, g, dox);
            // p = exp(s - m) / l = exp2(s log2(e) - (m log2(e) + log2(l)))
            if (g == 0) *reinterpret_cast<f32x4*>(&sm.stat[slot * 4]) = f32x4{m * LOG2E - __builtin_amdgcn_logf(inv_l), 0.f, DX, __uint_as_float(dbase)};
            __syncthreads();
            {
                f32x16 dk = zero16(), dv = zero16();
#pragma unroll
                for (int qblk = 0; qblk < NW; ++qblk) {
                    f32x16 sq = zero16(), dp = zero16();
                    rows_times_x(sm.rowA, base0 + 32 * qblk, tok, g, kx, sq);
                    rows_times_x(sm.rowB, base0 + 32 * qblk, tok, g, vx, dp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qs = base0 + 32 * qblk + key_of_reg(r, g);      // query slot
                        int same, qbb;
                        smap.decode(base0, qblk, key_of_reg(r, g), pl, same, qbb);
                        const bool ok = same && (qbb < S) && (p < HW);
                        const f32x4 stt = *reinterpret_cast<const f32x4*>(&sm.stat[qs * 4]);
                        const float pij = ok ? fast_exp_m(sq[r], stt[0]) : 0.f;
                        float pt = pij, dpr = dp[r];
                        if (DROP) {
                            const bool kept = ksr_attn_keep(__float_as_uint(stt[3]), b, keep);
                            pt = kept ? pij * inv_keep : 0.f;
                            dpr = kept ? dpr * inv_keep : 0.f;
                        }
                        sq[r] = pij * (dpr - stt[2]);    // dS
                        dp[r] = pt;                       // probabilities after dropout
                    }
                    tr_times_acc(sm.rowA, base0 + 32 * qblk, lane, sq, dk);
                    tr_times_acc(sm.rowB, base0 + 32 * qblk, lane, dp, dv);
                }
                store_dims32<ST>(xb, a.dqkv, rm, 192, 64 + 32 * X, lane, dk, 1.f);
                store_dims32<ST>(xb, a.dqkv, rm, 192, 128 + 32 * X, lane, dv, 1.f);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) { qb[0][s] = qb[1][s]; kb_[0][s] = kb_[1][s]; vb[0][s] = vb[1][s]; dob[0][s] = dob[1][s]; }
            D[0] = D[1];
        }
    }
}

}  // namespace

int ksr_attn_bf16_fwd_launch(const AttnArgs& a, int nw, unsigned grid, hipStream_t stream) {
    const dim3 g(grid), block(256);
    const bool bs = ksr_prec() == KSR_BF16S;
    if (bs && ksr_attn_dma_supported(a.Sp)) return ksr_attn_dma_fwd_launch(a, stream);
#define KSR_ATTN_FWD(NWV)                                                                               \
    do {                                                                                                \
        if (bs) hipLaunchKernelGGL((attn_fwd_bf16_kernel<NWV, __bf16>), g, block, 0, stream, a);        \
        else hipLaunchKernelGGL((attn_fwd_bf16_kernel<NWV, float>), g, block, 0, stream, a);            \
    } while (0)
    switch (nw) {
        case 1: KSR_ATTN_FWD(1); break;
        case 2: KSR_ATTN_FWD(2); break;
        default: KSR_ATTN_FWD(4); break;
    }
#undef KSR_ATTN_FWD
    KSR_LAUNCH_CHECK("batch_attn_fwd(bf16)");
    return KSR_OK;
}

int ksr_attn_bf16_bwd_launch(const AttnBwdArgs& a, int nw, unsigned grid, hipStream_t stream) {
    const dim3 g(grid), block(256);
    const bool drop = a.p_drop > 0.f;
    const bool bs = ksr_prec() == KSR_BF16S;
    if (bs && ksr_attn_dma_supported(a.Sp)) return ksr_attn_dma_bwd_launch(a, stream);
#define KSR_ATTN_BWD(NWV)                                                                                          \
    do {                                                                                                           \
        if (bs) {                                                                                                  \
            if (drop) hipLaunchKernelGGL((attn_bwd_bf16_kernel<NWV, true, __bf16>), g, block, 0, stream, a);       \
            else hipLaunchKernelGGL((attn_bwd_bf16_kernel<NWV, false, __bf16>), g, block, 0, stream, a);           \
        } else {                                                                                                   \
            if (drop) hipLaunchKernelGGL((attn_bwd_bf16_kernel<NWV, true, float>), g, block, 0, stream, a);        \
            else hipLaunchKernelGGL((attn_bwd_bf16_kernel<NWV, false, float>), g, block, 0, stream, a);            \
        }                                                                                                          \
    } while (0)
    switch (nw) {
        case 1: KSR_ATTN_BWD(1); break;
        case 2: KSR_ATTN_BWD(2); break;
        default: KSR_ATTN_BWD(4); break;
    }
#undef KSR_ATTN_BWD
    KSR_LAUNCH_CHECK("batch_attn_bwd(bf16)");
    return KSR_OK;
}
