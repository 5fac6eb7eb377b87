// Backward "data gradient" kernels of the separable-conv trunk and of the QKV projection:
//     [prologue that assembles dZ per pixel]  ->  dX = dZ * W (transposed 1x1 conv, MFMA)  ->  epilogue
// Same one-pixel-per-lane scheme as the forward (ksr_gemm.h).  The prologue may apply the
// TRANSPOSED depthwise 3x3 to a gradient tensor staged as an LDS halo tile (flipped taps), the
// epilogue applies the derivative of whatever preceded the 1x1 conv in the forward pass.
#include "ksr_conv_common.h"
#include "ksr_wg_fuse.h"

namespace {

enum { BPRO_PLAIN = 0, BPRO_CONVT = 1, BPRO_CONVT_G = 2 };
enum { BEPI_STORE = 0, BEPI_RELUMASK = 1, BEPI_SIGGRAD = 2, BEPI_GATE = 3, BEPI_RESID = 4, BEPI_RESID_SIGGRAD = 5,
       BEPI_STEM = 6 };

struct BwdArgs {
    const float* dz; int dz_ld;                 // [N, dz_ld] gradient of the 1x1 conv output (PLAIN: KIN read;
                                                //  CONVT_G: channels 0..63 read)
    float* dz_store; int dz_store_ld, dz_store_c0;   // CONVT*: the conv-transposed part is also stored here
    const float* halo_src;                      // [N,64] gradient to conv-transpose
    const float* dw_w;                          // [64][9] depthwise weights
    const float* mask_src; int mask_ld, mask_c0;     // CONVT_G: result *= (mask_src > 0)
    const float* wt;                            // [KIN][64] weights of the forward 1x1 conv (C_out = KIN)
    const float* e1; int e1_ld, e1_c0;
    const float* e2; int e2_ld, e2_c0;
    const float* resid;                         // [N,64]
    float* out1; int out1_ld, out1_c0;
    float* out2;                                // [N,64]
    const float* v; const float* p2w; float* dpre;   // STEM epilogue
    float* sums_slab;                           // STEM, optional: [3][gridDim.x][2][64] per-workgroup column sums
    float* wg_slab;                             // WGF: [gridDim.x][KSR_WGF_ROW] partial weight / bias gradients of the 1x1 conv
    const float* aff_v; const float* aff_ab;    // AFF: the 128-channel tensor t (e2 / mask_src) is relu(A[c] v[n] + B[c]): v [N], A|B [256]
    int B, H, W;
};

template <int PREC, int KIN, int BPRO>
struct __attribute__((aligned(16))) BwdSmem {
    float frag[2 * (KIN / 2) * 64 / (PREC != KSR_F32 ? 2 : 1)];
    float dww[BPRO == BPRO_PLAIN ? 4 : 9 * 64];
    float p2w[64];
    float aff[256];
    float halo[BPRO == BPRO_PLAIN ? 4 : HALO_FLOATS];
    // per-wave LDS transposers of the coalesced loads / stores (the stencil variants reuse the halo tile)
    float xpose[BPRO == BPRO_PLAIN ? 4 * KSR_XPOSE_FLOATS : 4];
};
static_assert(4 * KSR_XPOSE_FLOATS <= HALO_FLOATS, "transposers must fit in the halo tile");

// WGF: the kernel also accumulates the weight and bias gradient of the forward 1x1 conv whose output gradient is dz
// (KIN = 64; the conv's input is an operand of the epilogue: value * gate, the ReLU operand, or the sigmoid output).
// AFF: the saved 128-channel tensor of the first gated block is not read but rebuilt from one scalar per pixel,
// t[n][c] = relu(A[c] v[n] + B[c]) (see stem_gate_fwd_kernel): the value half in the GATE epilogue, the sign of the
// gate half as the CONVT_G mask.
// WGS: the same weight / bias gradient for the stencil instance <64, CONVT, SIGGRAD> (ffn3.pointwise1: dz = the
// conv-transposed gradient, input = the sigmoid output the epilogue reads), whose registers cannot take the 64 accumulators
// of the whole product: every wave writes its 32 pixels' operands as swizzled bf16 tiles into its slice of the (by then
// free) halo area, and after a barrier wave w accumulates the 32 x 32 block (w >> 1, w & 1) over all 128 pixels of the tile
// (16 accumulators, 8 matrix instructions).  dz_store is not written then.
template <int PREC, int KIN, int BPRO, int BEPI, bool WGF = false, bool AFF = false, bool WGS = false>
__global__ void __launch_bounds__(256, 2) dgrad_kernel(BwdArgs a) {
    static_assert(!WGS || (KIN == 64 && PREC == KSR_BF16S && BPRO == BPRO_CONVT && BEPI == BEPI_SIGGRAD && !WGF),
                  "weight gradient split over the waves: the <64, CONVT, SIGGRAD> instance in bf16 storage mode");
    static_assert(!AFF || BEPI == BEPI_GATE || BPRO == BPRO_CONVT_G, "AFF replaces e2 (GATE) or mask_src (CONVT_G)");
    static_assert(!WGF || (KIN == 64 && PREC == KSR_BF16S && BPRO == BPRO_PLAIN && (BEPI == BEPI_GATE || BEPI == BEPI_RELUMASK)),
                  "fused weight gradient: 64 -> 64 convs, bf16 storage mode, no stencil prologue (register budget)");
    constexpr int KH = KIN / 2;
    using ST = typename KsrStore<PREC>::T;
    __shared__ BwdSmem<PREC, KIN, BPRO> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    // the fused-weight-gradient instances sit at the register limit: with the 16-wide staging prologue the allocator
    // placed their spills in the tile loop (gate_pw_bwd_v 0.131 -> 0.158 ms), so they keep the narrow one
    ksr_stage_weights<PREC, KH, 2, true, WGF ? 1 : 16>(sm.frag, nullptr, a.wt, 64, nullptr, MapHL{32}, MapHL{KH}, tid, 256);
    if (BPRO != BPRO_PLAIN) ksr_stage_dw<true>(sm.dww, a.dw_w, tid);
    if (BEPI == BEPI_STEM && tid < 64) sm.p2w[tid] = a.p2w[tid];
    if (AFF) sm.aff[tid] = a.aff_ab[tid];
    __syncthreads();

    const int HW = a.H * a.W;
    const long N = (long)a.B * HW;
    constexpr bool spatial = (BPRO != BPRO_PLAIN);
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = spatial ? (long)a.B * tiles_y * tiles_x : (N + 127) / 128;

    // PLAIN: dz and the epilogue operands of a tile are fetched together, one memory latency for the whole tile; with
    // bf16 storage (tiles of 16 registers) they are requested one tile ahead: dz right after the previous tile's dz
    // has been unpacked, the epilogue operands after the previous epilogue has consumed its own (wide inputs only)
    constexpr bool PRE_R = BPRO == BPRO_PLAIN && (BEPI == BEPI_RESID || BEPI == BEPI_RESID_SIGGRAD);
    constexpr bool PRE_E1 = BPRO == BPRO_PLAIN && (BEPI == BEPI_RELUMASK || BEPI == BEPI_SIGGRAD || BEPI == BEPI_RESID_SIGGRAD || BEPI == BEPI_GATE);
    constexpr bool PRE_E2 = BPRO == BPRO_PLAIN && BEPI == BEPI_GATE && !AFF;
    constexpr bool AHEAD = BPRO == BPRO_PLAIN && sizeof(ST) == 2 && KIN > 64;     // the 64-wide instances keep 3 waves per SIMD without it
    KsrTile<ST> f[KIN / 64], pf_r, pf_e1, pf_e2;
    auto fetch_dz = [&](long t) {
        const long r0 = t * 128 + wv * 32;
        const int nv = (N - r0) < 32 ? (int)(N - r0 > 0 ? N - r0 : 0) : 32;
#pragma unroll
        for (int grp = 0; grp < KIN / 64; ++grp) ksr_co_fetch<ST, AHEAD>(a.dz, r0, nv, a.dz_ld, 64 * grp, lane, f[grp]);
    };
    auto fetch_epi = [&](long t) {
        const long r0 = t * 128 + wv * 32;
        const int nv = (N - r0) < 32 ? (int)(N - r0 > 0 ? N - r0 : 0) : 32;
        if constexpr (PRE_R) ksr_co_fetch<ST, AHEAD>(a.resid, r0, nv, 64, 0, lane, pf_r);
        if constexpr (PRE_E1) ksr_co_fetch<ST, AHEAD>(a.e1, r0, nv, a.e1_ld, a.e1_c0, lane, pf_e1);
        if constexpr (PRE_E2) ksr_co_fetch<ST, AHEAD>(a.e2, r0, nv, a.e2_ld, a.e2_c0, lane, pf_e2);
    };
    if (AHEAD && (long)blockIdx.x < ntiles) { fetch_dz(blockIdx.x); fetch_epi(blockIdx.x); }
    // stencil prologues with bf16 storage: the next halo tile is requested while this one is processed
    constexpr bool HAHEAD = spatial && sizeof(ST) == 2;
    KsrHalo<ST> hr;
    auto fetch_halo = [&](long t) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const int tb = (int)(t / ((long)tiles_x * tiles_y));
        ksr_halo_fetch<ST>(a.halo_src, 64, 0, tb, ty * 4, tx * 32, a.H, a.W, tid, hr);
    };
    if (HAHEAD && (long)blockIdx.x < ntiles) fetch_halo(blockIdx.x);

    // STEM with sums_slab: sum_n d[n,c] * v[n] and sum_n d[n,c] for d = the two halves of dz and the output (lane = c)
    float cs[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const bool col_sums = BEPI == BEPI_STEM && a.sums_slab != nullptr;
    KsrWgFused wgf;
    if constexpr (WGF) ksr_wgf_init(wgf);
    f32x16 wgs_acc;          // WGS: block (wv >> 1, wv & 1) of dW over every pixel of the workgroup's tiles
    float wgs_bias = 0.f;    // WGS, waves 0 and 2: sum of dz[.][32 (wv >> 1) + (lane & 31)] over the pixels of waves 2 (lane >> 5), + 1
#pragma unroll
    for (int q = 0; q < 16; ++q) wgs_acc[q] = 0.f;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        bool live;
        long n, n0w;          // this lane's pixel; first pixel of the wave's 32-pixel run
        int nvalid;           // real pixels of that run
        if (spatial) {
            const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
            const int b = (int)(tile / ((long)tiles_x * tiles_y));
            const int y0 = ty * 4, x0 = tx * 32;
            const int y = y0 + wv, x = x0 + tok;
            live = (y < a.H) && (x < a.W);
            n = live ? (long)b * HW + (long)y * a.W + x : 0;
            n0w = (long)b * HW + (long)y * a.W + x0;
            nvalid = (y < a.H) ? ((a.W - x0) < 32 ? a.W - x0 : 32) : 0;
            __syncthreads();
            if (!HAHEAD) fetch_halo(tile);
            ksr_halo_commit<ST>(sm.halo, tid, hr);
            __syncthreads();
            if (HAHEAD && tile + gridDim.x < ntiles) fetch_halo(tile + gridDim.x);     // in flight during this tile
        } else {
            n = tile * 128 + wv * 32 + tok;
            live = n < N;
            if (!live) n = 0;
            n0w = tile * 128 + wv * 32;
            nvalid = (N - n0w) < 32 ? (int)(N - n0w > 0 ? N - n0w : 0) : 32;
        }
        float* xb = (spatial ? sm.halo : sm.xpose) + wv * KSR_XPOSE_FLOATS;
        // AFF: this pixel's scalar, requested before the tile's other work so that its latency is covered
        float aff_vv = 0.f;
        if constexpr (AFF) aff_vv = live ? a.aff_v[n] : 0.f;

        float dz[KH];
        if (BPRO == BPRO_PLAIN) {
            if (!AHEAD) { fetch_dz(tile); fetch_epi(tile); }
#pragma unroll
            for (int grp = 0; grp < KIN / 64; ++grp) {
                float t[32];
                ksr_co_finish<ST>(xb, lane, f[grp], t);
#pragma unroll
                for (int i = 0; i < 32; ++i) dz[32 * grp + i] = t[i];
            }
            if (AHEAD && tile + gridDim.x < ntiles) fetch_dz(tile + gridDim.x);
        } else {
            float g[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) g[i] = 0.f;
            ksr_dwconv_lane(sm.halo, sm.dww, wv, tok, h, g);
            __syncthreads();       // every wave is done with the halo tile: it now serves as the transposers
            if (BPRO == BPRO_CONVT_G) {
                float mk[32], t[32];
                if constexpr (AFF) {
                    const float vv = aff_vv;
#pragma unroll
                    for (int i = 0; i < 32; ++i) mk[i] = live ? fmaf(sm.aff[64 + 32 * h + i], vv, sm.aff[192 + 32 * h + i]) : 0.f;   // t is 0 outside the image
                } else {
                    ksr_load32_co<ST>(xb, a.mask_src, n0w, nvalid, a.mask_ld, a.mask_c0, lane, mk);
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) g[i] = mk[i] > 0.f ? g[i] : 0.f;
                ksr_load32_co<ST>(xb, a.dz, n0w, nvalid, a.dz_ld, 0, lane, t);
#pragma unroll
                for (int i = 0; i < 32; ++i) { dz[i] = t[i]; dz[32 + i] = g[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) dz[i] = g[i];
            }
        }

        float vs = 0.f;
        if (BEPI == BEPI_STEM && col_sums) {
            vs = live ? a.v[n] : 0.f;
            if constexpr (KIN == 128) {
                float d0[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) d0[i] = dz[i];
                ksr_colsum32_scalar(xb, lane, d0, vs, cs[0][0], cs[0][1]);
#pragma unroll
                for (int i = 0; i < 32; ++i) d0[i] = dz[32 + i];
                ksr_colsum32_scalar(xb, lane, d0, vs, cs[1][0], cs[1][1]);
            }
        }
        // phase fences: keep the epilogue's loads from being hoisted above the GEMM (register pressure)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2];
        ksr_wave_gemm<PREC, KH, 2>(sm.frag, nullptr, dz, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (BPRO != BPRO_PLAIN) {
            // the conv-transposed gradient is also an output
            float t[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) t[i] = dz[(BPRO == BPRO_CONVT_G ? 32 : 0) + i];
            if (a.dz_store) ksr_store32_co<ST>(xb, a.dz_store, n0w, nvalid, a.dz_store_ld, a.dz_store_c0, lane, t);
            __builtin_amdgcn_sched_barrier(0);
        }
        float dA[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) { dA[i] = acc[0][i]; dA[16 + i] = acc[1][i]; }

        if (BEPI == BEPI_RESID || BEPI == BEPI_RESID_SIGGRAD) {
            float r[32];
            if constexpr (PRE_R) ksr_co_finish<ST>(xb, lane, pf_r, r);
            else ksr_load32_co<ST>(xb, a.resid, n0w, nvalid, 64, 0, lane, r);
#pragma unroll
            for (int i = 0; i < 32; ++i) dA[i] += r[i];
        }
        if (BEPI == BEPI_RELUMASK) {
            float e[32];
            if constexpr (PRE_E1) ksr_co_finish<ST>(xb, lane, pf_e1, e);
            else ksr_load32_co<ST>(xb, a.e1, n0w, nvalid, a.e1_ld, a.e1_c0, lane, e);
#pragma unroll
            for (int i = 0; i < 32; ++i) dA[i] = e[i] > 0.f ? dA[i] : 0.f;
            if constexpr (WGF) {
                float d0[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) d0[i] = dz[i];
                ksr_wgf_accumulate(xb, lane, d0, e, wgf);
            }
        }
        uint4 wgs_e[4];          // WGS: the conv's input (the sigmoid output) as bf16, for the tile written below
        if (BEPI == BEPI_SIGGRAD || BEPI == BEPI_RESID_SIGGRAD) {
            float e[32];
            if constexpr (PRE_E1) ksr_co_finish<ST>(xb, lane, pf_e1, e);
            else ksr_load32_co<ST>(xb, a.e1, n0w, nvalid, a.e1_ld, a.e1_c0, lane, e);
#pragma unroll
            for (int i = 0; i < 32; ++i) dA[i] = dA[i] * e[i] * (1.f - e[i]);
            if constexpr (WGS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) wgs_e[q] = ksr_pack8(e + 8 * q);
            }
        }
        if (BEPI == BEPI_GATE) {
            float gte[32], val[32], o2[32];
            if constexpr (PRE_E1) ksr_co_finish<ST>(xb, lane, pf_e1, gte);
            else ksr_load32_co<ST>(xb, a.e1, n0w, nvalid, a.e1_ld, a.e1_c0, lane, gte);
            if constexpr (AFF) {
                const float vv = aff_vv;
#pragma unroll
                for (int i = 0; i < 32; ++i) val[i] = live ? fmaxf(fmaf(sm.aff[32 * h + i], vv, sm.aff[128 + 32 * h + i]), 0.f) : 0.f;
            } else if constexpr (PRE_E2) ksr_co_finish<ST>(xb, lane, pf_e2, val);
            else ksr_load32_co<ST>(xb, a.e2, n0w, nvalid, a.e2_ld, a.e2_c0, lane, val);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                o2[i] = dA[i] * val[i];
                dA[i] = val[i] > 0.f ? dA[i] * gte[i] : 0.f;
            }
            ksr_store32_co<ST, !AHEAD>(xb, a.out2, n0w, nvalid, 64, 0, lane, o2);
            if constexpr (WGF) {
                float d0[32], m[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) { d0[i] = dz[i]; m[i] = val[i] * gte[i]; }
                ksr_wgf_accumulate(xb, lane, d0, m, wgf);
            }
        }
        if (BEPI == BEPI_STEM) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) s += dA[i] * sm.p2w[32 * h + i];
            s = ksr_pair_sum(s);
            if (live && h == 0) a.dpre[n] = a.v[n] > 0.f ? s : 0.f;
            if (col_sums) ksr_colsum32_scalar(xb, lane, dA, vs, cs[2][0], cs[2][1]);
        }
        if (a.out1) ksr_store32_co<ST, !AHEAD>(xb, a.out1, n0w, nvalid, a.out1_ld, a.out1_c0, lane, dA);
        if (AHEAD && tile + gridDim.x < ntiles) fetch_epi(tile + gridDim.x);
        if constexpr (WGS) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            union Op { s16x4 hv[2]; bf16x8 v; };
            __bf16* tz = reinterpret_cast<__bf16*>(xb);
            __bf16* ta = tz + 32 * 64;
            ksr_wave_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float d8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) d8[j] = live ? dz[8 * q + j] : 0.f;      // pixels outside the image have no gradient
                *reinterpret_cast<uint4*>(tz + ksr_wgf_off(tok, 4 * h + q)) = ksr_pack8(d8);
                *reinterpret_cast<uint4*>(ta + ksr_wgf_off(tok, 4 * h + q)) = wgs_e[q];
            }
            __syncthreads();
            const int g5 = lane >> 5, db = (lane >> 4) & 1, qq = (lane & 15) >> 2, p4 = lane & 3;
            const int c2 = 2 * db + (p4 >> 1), jb = wv >> 1, kb = wv & 1;
            auto tr = [&](const __bf16* tile, int row, int chunk) -> s16x4 {
                return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(tile + ksr_wgf_off(row, chunk) + 4 * (p4 & 1)));
            };
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const __bf16* base = reinterpret_cast<const __bf16*>(sm.halo + w * KSR_XPOSE_FLOATS);
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int r0 = 16 * st + 4 * g5 + qq;
                    Op A, Bv;
                    A.hv[0] = tr(base, r0, 4 * jb + c2); A.hv[1] = tr(base, r0 + 8, 4 * jb + c2);
                    Bv.hv[0] = tr(base + 2048, r0, 4 * kb + c2); Bv.hv[1] = tr(base + 2048, r0 + 8, 4 * kb + c2);
                    wgs_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, Bv.v, wgs_acc, 0, 0, 0);
                }
            }
            if (kb == 0) {
                const int ch = 32 * jb + (lane & 31);
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const __bf16* base = reinterpret_cast<const __bf16*>(sm.halo + (2 * g5 + w2) * KSR_XPOSE_FLOATS);
#pragma unroll
                    for (int r = 0; r < 32; r += 2) {
                        s0 += (float)base[ksr_wgf_off(r, ch >> 3) + (ch & 7)];
                        s1 += (float)base[ksr_wgf_off(r + 1, ch >> 3) + (ch & 7)];
                    }
                }
                wgs_bias += s0 + s1;
            }
            // the barrier at the top of the next tile (or the one below) ends this phase
        }
    }
    if constexpr (WGS) {
        __syncthreads();
        float* row = a.wg_slab + (long)blockIdx.x * KSR_WGF_ROW;
        const int g5 = lane >> 5, jb = wv >> 1, kb = wv & 1;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            row[(32 * jb + (q & 3) + 8 * (q >> 2) + 4 * g5) * 64 + 32 * kb + (lane & 31)] = wgs_acc[q];
        const float other = __shfl_xor(wgs_bias, 32, 64);
        if (kb == 0 && g5 == 0) row[64 * 64 + 32 * jb + (lane & 31)] = wgs_bias + other;
    }
    if constexpr (WGF) ksr_wgf_finish(spatial ? sm.halo : sm.xpose, a.wg_slab + (long)blockIdx.x * KSR_WGF_ROW, tid, wgf);
    if (BEPI == BEPI_STEM && col_sums) {
        // the four waves' sums meet in LDS (the halo / transposer area is free once every wave has left the loop)
        float* area = spatial ? sm.halo : sm.xpose;
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int q = 0; q < 2; ++q) area[wv * KSR_XPOSE_FLOATS + (2 * l + q) * 64 + lane] = cs[l][q];
        __syncthreads();
        for (int i = tid; i < 6 * 64; i += 256) {
            const float v = (area[i] + area[KSR_XPOSE_FLOATS + i]) + (area[2 * KSR_XPOSE_FLOATS + i] + area[3 * KSR_XPOSE_FLOATS + i]);
            const int l = i / 128, r = i - 128 * l;
            a.sums_slab[((long)l * gridDim.x + blockIdx.x) * 128 + r] = v;
        }
    }
}

// fn_1 backward of the first gated block without the matrix product.  The block's input is x1[n][k] = p2w[k] v[n] +
// p2b[k] (one scalar per pixel), so everything its backward must deliver is a function of dt [N,128] alone:
//     dpre[n]              = (v[n] > 0) * sum_c dt[n][c] * wt[c],   wt[c] = sum_k W[c][k] p2w[k]
//     sum_n dx[n][k] * v[n] = sum_c W[c][k] * (sum_n dt[n][c] v[n]),  likewise sum_n dx[n][k]
// (dx = dt * W is never formed).  dt's value half is read, its gate half is the transposed depthwise 3x3 of dgate times
// the sign of t's gate half, rebuilt from v (t = relu(A v + B)).  Every lane keeps the four running column sums of its
// own pixel's 32 channels over all tiles of the workgroup (128 registers); they cross lanes once, at the end.
struct StemBwdArgs {
    const void* dt; const void* dgate;          // [N,128] (channels 0..63 read), [N,64]
    const float* dw_w;                          // [64][9] gate weights
    const float* v; const float* ab;            // [N]; A|B [256]
    const float* fn1_w; const float* p2w;       // [128][64]; [64]
    float* dpre;                                // [N]
    float* sums_slab;                           // [3][gridDim.x][2][64]
    int B, H, W;
};

struct __attribute__((aligned(16))) StemBwdSmem {
    float dww[9 * 64];
    float affA[64], affB[64];
    float wt[128];
    float halo[HALO_FLOATS];
};

template <class ST>
__global__ void __launch_bounds__(256, 2) fn1_stem_bwd_kernel(StemBwdArgs a) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    __shared__ StemBwdSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_dw<true>(sm.dww, a.dw_w, tid);
    if (tid < 64) { sm.affA[tid] = a.ab[64 + tid]; sm.affB[tid] = a.ab[192 + tid]; }
    if (tid >= 128) {
        const int c = tid - 128;
        float s = 0.f;
        for (int k = 0; k < 64; ++k) s = fmaf(a.fn1_w[c * 64 + k], a.p2w[k], s);
        sm.wt[c] = s;
    }
    __syncthreads();

    const int HW = a.H * a.W;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    KsrHalo<ST> hr;
    auto fetch_halo = [&](long t) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const int tb = (int)(t / ((long)tiles_x * tiles_y));
        ksr_halo_fetch<ST>(a.dgate, 64, 0, tb, ty * 4, tx * 32, a.H, a.W, tid, hr);
    };
    const KsrTileWalk tw = ksr_xcd_tile_walk(ntiles);
    if (tw.first < tw.end) fetch_halo(tw.first);

    // Column sums over tokens on the matrix cores (the scheme of ksr_wg_fuse.h): the wave's 32 x 64 tile of d goes to
    // LDS as swizzled bf16 rows and comes back transposed as the A operand (channels x tokens); the B operand's columns
    // 0, 1, 2 hold, per token, the upper and the lower bf16 half of v and 1.  accV / accG [block of 32 channels]: lane
    // column (lane & 31) = 0, 1: sum d v (upper, lower part), 2: sum d, for 16 channels each.
    f32x16 accV[2], accG[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) { accV[0][q] = 0.f; accV[1][q] = 0.f; accG[0][q] = 0.f; accG[1][q] = 0.f; }
    const int g5 = lane >> 5, db = (lane >> 4) & 1, qq = (lane & 15) >> 2, p4 = lane & 3;
    auto tr = [&](const __bf16* tile, int row, int chunk) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + ksr_wgf_off(row, chunk) + 4 * (p4 & 1)));
    };
    union Op { s16x4 hv[2]; bf16x8 v; };

    for (long tile = tw.first; tile < tw.end; tile += tw.step) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const int b = (int)(tile / ((long)tiles_x * tiles_y));
        const int y0 = ty * 4, x0 = tx * 32;
        const int y = y0 + wv, x = x0 + tok;
        const bool live = (y < a.H) && (x < a.W);
        const long n = live ? (long)b * HW + (long)y * a.W + x : 0;
        const long n0w = (long)b * HW + (long)y * a.W + x0;
        const int nvalid = (y < a.H) ? ((a.W - x0) < 32 ? a.W - x0 : 32) : 0;
        __syncthreads();
        ksr_halo_commit<ST>(sm.halo, tid, hr);
        __syncthreads();
        if (tile + tw.step < tw.end) fetch_halo(tile + tw.step);     // in flight during this tile
        // this tile's value half and scalar: requested before the stencil, which covers their latency
        KsrTile<ST> tv;
        ksr_co_fetch<ST>(a.dt, n0w, nvalid, 128, 0, lane, tv);
        const float vv = live ? a.v[n] : 0.f;

        float g[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) g[i] = 0.f;
        ksr_dwconv_lane(sm.halo, sm.dww, wv, tok, h, g);
        __syncthreads();       // every wave is done with the halo tile: its first part now holds the waves' bf16 tiles
        __bf16* tz = reinterpret_cast<__bf16*>(sm.halo + wv * KSR_XPOSE_FLOATS);
        __bf16* ta = tz + 32 * 64;
        // value half: the fetched pieces go to the swizzled tile as they are (bf16 storage) or rounded (fp32 storage)
        if constexpr (sizeof(ST) == 2) {
            const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tz + ksr_wgf_off(8 * i + r8, c8)) = tv.v[i];
        } else {
            const int r4 = lane >> 4, c4 = lane & 15;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x4[4] = {tv.v[i].x, tv.v[i].y, tv.v[i].z, tv.v[i].w};
                bf16x8 pk;      // only the first four are stored
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (__bf16)x4[j];
                *reinterpret_cast<uint2*>(tz + ksr_wgf_off(4 * i + r4, c4 >> 1) + 4 * (c4 & 1)) = *reinterpret_cast<const uint2*>(&pk);
            }
        }
        if (h == 0) {
            const float hi = __uint_as_float(__float_as_uint(vv) & 0xffff0000u);
            const float x8[8] = {hi, vv - hi, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<uint4*>(ta + ksr_wgf_off(tok, 0)) = ksr_pack8(x8);
        }
        ksr_wave_fence();
        Op Bv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r0 = 16 * s + 4 * g5 + qq;
            const int chunk = 2 * db + (p4 >> 1);
            Bv[s].hv[0] = tr(ta, r0, chunk); Bv[s].hv[1] = tr(ta, r0 + 8, chunk);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                Op A;
                A.hv[0] = tr(tz, r0, 4 * blk + chunk); A.hv[1] = tr(tz, r0 + 8, 4 * blk + chunk);
                accV[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, Bv[s].v, accV[blk], 0, 0, 0);
            }
        }
        // this lane's 32 value-half channels, for the per-pixel product
        float s = 0.f;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x8[8];
                ksr_unpack8(*reinterpret_cast<const uint4*>(tz + ksr_wgf_off(tok, 4 * h + q)), x8);
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(x8[j], sm.wt[32 * h + 8 * q + j], s);
            }
        }
        // gate half: transposed depthwise conv of dgate, times the sign of t's gate half
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float tg = fmaf(sm.affA[32 * h + i], vv, sm.affB[32 * h + i]);
            g[i] = (live && tg > 0.f) ? g[i] : 0.f;          // t is 0 outside the image
            s = fmaf(g[i], sm.wt[64 + 32 * h + i], s);
        }
        ksr_wave_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(tz + ksr_wgf_off(tok, 4 * h + q)) = ksr_pack8(g + 8 * q);
        ksr_wave_fence();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int r0 = 16 * s2 + 4 * g5 + qq;
            const int chunk = 2 * db + (p4 >> 1);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                Op A;
                A.hv[0] = tr(tz, r0, 4 * blk + chunk); A.hv[1] = tr(tz, r0 + 8, 4 * blk + chunk);
                accG[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, Bv[s2].v, accG[blk], 0, 0, 0);
            }
        }
        s = ksr_pair_sum(s);
        if (live && h == 0) a.dpre[n] = vv > 0.f ? s : 0.f;
    }

    // the waves' sums meet in LDS: cols[wave][column 0..2][half l][channel c]
    __syncthreads();
    float* cols = sm.halo;                               // [4][3][128]
    float* S = sm.halo + 4 * 3 * 128;                    // [256]: (half l, q, c) at 128 l + 64 q + c
    float* part = S + 256;                               // [256]
    {
        const int col = lane & 31;
        if (col < 3) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = 32 * blk + (q & 3) + 8 * (q >> 2) + 4 * g5;
                    cols[(wv * 3 + col) * 128 + c] = accV[blk][q];
                    cols[(wv * 3 + col) * 128 + 64 + c] = accG[blk][q];
                }
        }
    }
    __syncthreads();
    {
        const int l = tid >> 7, q = (tid >> 6) & 1, c = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* cw = cols + w * 3 * 128 + 64 * l + c;
            t += q == 0 ? cw[0] + cw[128] : cw[256];
        }
        S[tid] = t;
        a.sums_slab[((long)l * gridDim.x + blockIdx.x) * 128 + (tid & 127)] = t;
    }
    __syncthreads();
    {   // this workgroup's share of sum_n dx[n][k] * v[n] (q = 0) and sum_n dx[n][k] (q = 1)
        const int k = tid & 63, q = (tid >> 6) & 1, l = tid >> 7;
        float p = 0.f;
        for (int c = 0; c < 64; ++c) p = fmaf(a.fn1_w[(64 * l + c) * 64 + k], S[128 * l + 64 * q + c], p);
        part[tid] = p;
    }
    __syncthreads();
    if (tid < 128) a.sums_slab[((long)2 * gridDim.x + blockIdx.x) * 128 + tid] = part[tid] + part[tid + 128];
}
static_assert(4 * KSR_XPOSE_FLOATS + 512 <= HALO_FLOATS, "the final sums must fit in the halo tile");

// out[l][q * stride[l] + c] = sum over workgroups of slab[l][g][q][c]  (l < 3, q < 2, c < 64), fixed order
struct ColRedArgs { const float* slab; int G; float* out[3]; int stride[3]; };
__global__ void __launch_bounds__(1024) colsum_reduce_kernel(ColRedArgs a) {
    // grid (3, 4): 32 of the 128 sums per workgroup x 32 slab groups, four independent partial sums per thread
    __shared__ float part[32][33];
    const int l = blockIdx.x;
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
    a.out[l][(e >> 6) * a.stride[l] + (e & 63)] = s;
}

// Stencil (spatial) kernels: at most 512 workgroups (their halo tiles allow two per CU; the stem's column-sum reduction
// relies on this count).  Plain kernels: one round of persistent workgroups, at most 768 (the fused weight gradient's slab).
unsigned dgrad_grid_spatial(const BwdArgs& a) {
    const long ntiles = (long)a.B * ((a.H + 3) / 4) * ((a.W + 31) / 32);
    return (unsigned)(ntiles < 512 ? ntiles : 512);
}
template <class K>
unsigned dgrad_grid(K kernel, const BwdArgs& a, bool spatial) {
    if (spatial) return dgrad_grid_spatial(a);
    const long N = (long)a.B * a.H * a.W;
    return ksr_persistent_grid(kernel, (N + 127) / 128, 768);
}

template <int PREC, int KIN, int BPRO, int BEPI, bool AFF = false>
int launch_dgrad_p(const BwdArgs& a, hipStream_t s) {
    const long N = (long)a.B * a.H * a.W;
    const bool spatial = BPRO != BPRO_PLAIN;
    const long ntiles = spatial ? (long)a.B * ((a.H + 3) / 4) * ((a.W + 31) / 32) : (N + 127) / 128;
    (void)ntiles;
    hipLaunchKernelGGL((dgrad_kernel<PREC, KIN, BPRO, BEPI, false, AFF>),
                       dim3(dgrad_grid(dgrad_kernel<PREC, KIN, BPRO, BEPI, false, AFF>, a, spatial)), dim3(256), 0, s, a);
    KSR_LAUNCH_CHECK("dgrad");
    return KSR_OK;
}

// Data gradient + fused weight gradient of a 64 -> 64 conv (bf16 storage mode), then the reduction of the partial rows.
template <int BPRO, int BEPI, bool AFF = false>
int launch_dgrad_wgf(const BwdArgs& a, float* dW, float* db, hipStream_t s) {
    KSR_CHECK_ARG(a.wg_slab && dW && db, "fused weight gradient: null pointer");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "fused weight gradient: bf16 storage mode only (the other modes use ksr_wgrad)");
    const unsigned grid = dgrad_grid(dgrad_kernel<KSR_BF16S, 64, BPRO, BEPI, true, AFF>, a, BPRO != BPRO_PLAIN);
    hipLaunchKernelGGL((dgrad_kernel<KSR_BF16S, 64, BPRO, BEPI, true, AFF>), dim3(grid), dim3(256), 0, s, a);
    KSR_LAUNCH_CHECK("dgrad(fused weight gradient)");
    return ksr_slab_reduce(a.wg_slab, (int)grid, 64, 64, dW, db, s);
}

// <64, CONVT, SIGGRAD> with the 1x1 conv's weight gradient split over the waves, then the reduction of the partial rows.
int launch_dgrad_wgs(const BwdArgs& a, float* dW, float* db, hipStream_t s) {
    KSR_CHECK_ARG(a.wg_slab && dW && db, "fused weight gradient: null pointer");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "fused weight gradient: bf16 storage mode only (the other modes use ksr_wgrad)");
    const unsigned grid = dgrad_grid(dgrad_kernel<KSR_BF16S, 64, BPRO_CONVT, BEPI_SIGGRAD, false, false, true>, a, true);
    hipLaunchKernelGGL((dgrad_kernel<KSR_BF16S, 64, BPRO_CONVT, BEPI_SIGGRAD, false, false, true>), dim3(grid), dim3(256), 0, s, a);
    KSR_LAUNCH_CHECK("dgrad(weight gradient split over the waves)");
    return ksr_slab_reduce(a.wg_slab, (int)grid, 64, 64, dW, db, s);
}

template <int KIN, int BPRO, int BEPI, bool AFF = false>
int launch_dgrad(const BwdArgs& a, hipStream_t s) {
#define KSR_CALL(P) return launch_dgrad_p<P, KIN, BPRO, BEPI, AFF>(a, s)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    return KSR_OK;
}

int dims_ok(int B, int H, int W, const char* who) {
    KSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "%s: bad dimensions", who);
    return KSR_OK;
}

}  // namespace

// GatedFFN tail backward (fn_2 + value*gate product), used for ffn2 and ffn4.
extern "C" int ksr_gate_pw_bwd(const float* dz, const float* fn2_w, const float* gate_saved, const float* t_saved,
                               float* dt_value, float* dgate, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dz && fn2_w && gate_saved && t_saved && dt_value && dgate, "gate_pw_bwd: null pointer");
    if (int rc = dims_ok(B, H, W, "gate_pw_bwd")) return rc;
    BwdArgs a{};
    a.dz = dz; a.dz_ld = 64; a.wt = fn2_w; a.e1 = gate_saved; a.e1_ld = 64; a.e2 = t_saved; a.e2_ld = 128;
    a.out1 = dt_value; a.out1_ld = 128; a.out1_c0 = 0; a.out2 = dgate; a.B = B; a.H = H; a.W = W;
    return launch_dgrad<64, BPRO_PLAIN, BEPI_GATE>(a, stream);
}

// fn_1 (64 -> 128) backward: completes dt[:, 64:128] = convT(dgate) * (t[:, 64:] > 0) and computes
// dx = dt * W.  stem = 1: the input of fn_1 is the first FFN (1 -> 64): also emits dpre [N].
// stem with sums_slab (3 * 512 * 128 floats of workspace): the kernel also reduces, over all pixels,
//   dt_sums[0][c] = sum dt[n,c] * v[n], dt_sums[1][c] = sum dt[n,c]   (c < 128, rows 128 floats apart) and
//   dx_sums[c] = sum dx[n,c] * v[n], dx_sums[dx_row_stride + c] = sum dx[n,c]   (c < 64),
// which is all the stem's parameter gradients need of dt[:, 64:128] and dx: neither is written then (dx may be NULL).
extern "C" int ksr_fn1_bwd(float* dt, const float* dgate, const float* gate_w, const float* t_saved,
                           const float* fn1_w, float* dx, int stem, const float* v_saved, const float* pw2_w,
                           float* dpre, float* sums_slab, float* dt_sums, float* dx_sums, int dx_row_stride,
                           int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dt && dgate && gate_w && t_saved && fn1_w, "fn1_bwd: null pointer");
    if (int rc = dims_ok(B, H, W, "fn1_bwd")) return rc;
    BwdArgs a{};
    a.dz = dt; a.dz_ld = 128; a.dz_store = dt; a.dz_store_ld = 128; a.dz_store_c0 = 64;
    a.halo_src = dgate; a.dw_w = gate_w; a.mask_src = t_saved; a.mask_ld = 128; a.mask_c0 = 64;
    a.wt = fn1_w; a.out1 = dx; a.out1_ld = 64; a.B = B; a.H = H; a.W = W;
    if (stem) {
        KSR_CHECK_ARG(v_saved && pw2_w && dpre, "fn1_bwd(stem): null pointer");
        a.v = v_saved; a.p2w = pw2_w; a.dpre = dpre;
        if (sums_slab) {
            KSR_CHECK_ARG(dt_sums && dx_sums && dx_row_stride >= 64, "fn1_bwd(stem): sums_slab needs dt_sums, dx_sums and a row stride >= 64");
            a.sums_slab = sums_slab;
            a.dz_store = nullptr;
            a.out1 = nullptr;
        } else {
            KSR_CHECK_ARG(dx, "fn1_bwd: null pointer");
        }
        if (int rc = launch_dgrad<128, BPRO_CONVT_G, BEPI_STEM>(a, stream)) return rc;
        if (sums_slab) {
            ColRedArgs r{sums_slab, (int)dgrad_grid_spatial(a), {dt_sums, dt_sums + 64, dx_sums}, {128, 128, dx_row_stride}};
            hipLaunchKernelGGL(colsum_reduce_kernel, dim3(3, 4), dim3(1024), 0, stream, r);
            KSR_LAUNCH_CHECK("fn1_bwd(column sums)");
        }
        return KSR_OK;
    }
    KSR_CHECK_ARG(dx, "fn1_bwd: null pointer");
    return launch_dgrad<128, BPRO_CONVT_G, BEPI_STORE>(a, stream);
}

// FFN pointwise2 backward: ds = (dy * W) * (operand > 0), operand = relu(dw(.)) saved by the forward.
extern "C" int ksr_pw2_bwd(const float* dy, const float* pw2_w, const float* operand_saved, float* ds, int B, int H,
                           int W, hipStream_t stream) {
    KSR_CHECK_ARG(dy && pw2_w && operand_saved && ds, "pw2_bwd: null pointer");
    if (int rc = dims_ok(B, H, W, "pw2_bwd")) return rc;
    BwdArgs a{};
    a.dz = dy; a.dz_ld = 64; a.wt = pw2_w; a.e1 = operand_saved; a.e1_ld = 64; a.out1 = ds; a.out1_ld = 64;
    a.B = B; a.H = H; a.W = W;
    return launch_dgrad<64, BPRO_PLAIN, BEPI_RELUMASK>(a, stream);
}

// FFN pointwise1 backward: dt = convT(ds) (stored), dx = dt * W, then the sigmoid that fed the FFN:
// dz_prev = dx * y * (1 - y).
extern "C" int ksr_pw1_bwd(const float* ds, const float* dw_w, const float* pw1_w, const float* y_saved, float* dt,
                           float* dz_prev, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(ds && dw_w && pw1_w && y_saved && dt && dz_prev, "pw1_bwd: null pointer");
    if (int rc = dims_ok(B, H, W, "pw1_bwd")) return rc;
    BwdArgs a{};
    a.halo_src = ds; a.dw_w = dw_w; a.dz_store = dt; a.dz_store_ld = 64; a.dz_store_c0 = 0; a.wt = pw1_w;
    a.e1 = y_saved; a.e1_ld = 64; a.out1 = dz_prev; a.out1_ld = 64; a.B = B; a.H = H; a.W = W;
    return launch_dgrad<64, BPRO_CONVT, BEPI_SIGGRAD>(a, stream);
}

// ksr_pw1_bwd with pointwise1's weight / bias gradient (dt^T y_saved) accumulated in the same pass (bf16 storage mode): dt,
// which only fed that product, is not written.  slab: ksr_fused_wgrad_slab_floats() floats of workspace.
extern "C" int ksr_pw1_bwd_wg(const float* ds, const float* dw_w, const float* pw1_w, const float* y_saved, float* dz_prev,
                              float* slab, float* dW, float* db, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(ds && dw_w && pw1_w && y_saved && dz_prev, "pw1_bwd_wg: null pointer");
    if (int rc = dims_ok(B, H, W, "pw1_bwd_wg")) return rc;
    BwdArgs a{};
    a.halo_src = ds; a.dw_w = dw_w; a.wt = pw1_w; a.e1 = y_saved; a.e1_ld = 64; a.out1 = dz_prev; a.out1_ld = 64;
    a.wg_slab = slab; a.B = B; a.H = H; a.W = W;
    return launch_dgrad_wgs(a, dW, db, stream);
}

// QKV projection backward + residual: ds = resid + dqkv * W_in, optionally followed by the sigmoid
// derivative of the tensor that entered the first transformer layer (y_saved != NULL).
extern "C" int ksr_in_proj_bwd(const float* dqkv, const float* in_w, const float* resid, const float* y_saved,
                               float* ds, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dqkv && in_w && resid && ds, "in_proj_bwd: null pointer");
    if (int rc = dims_ok(B, H, W, "in_proj_bwd")) return rc;
    BwdArgs a{};
    a.dz = dqkv; a.dz_ld = 192; a.wt = in_w; a.resid = resid; a.e1 = y_saved; a.e1_ld = 64;
    a.out1 = ds; a.out1_ld = 64; a.B = B; a.H = H; a.W = W;
    if (y_saved) return launch_dgrad<192, BPRO_PLAIN, BEPI_RESID_SIGGRAD>(a, stream);
    return launch_dgrad<192, BPRO_PLAIN, BEPI_RESID>(a, stream);
}

// Plain 1x1-conv data gradient: dx [N,64] = dz [N,n_out] * W  (W [n_out][64], n_out 64 or 192).
extern "C" int ksr_pointwise_bwd(const float* dz, int n_out, const float* w, float* dx, int B, int H, int W,
                                 hipStream_t stream) {
    KSR_CHECK_ARG(dz && w && dx, "pointwise_bwd: null pointer");
    if (int rc = dims_ok(B, H, W, "pointwise_bwd")) return rc;
    BwdArgs a{};
    a.dz = dz; a.dz_ld = n_out; a.wt = w; a.out1 = dx; a.out1_ld = 64; a.B = B; a.H = H; a.W = W;
    if (n_out == 64) return launch_dgrad<64, BPRO_PLAIN, BEPI_STORE>(a, stream);
    if (n_out == 192) return launch_dgrad<192, BPRO_PLAIN, BEPI_STORE>(a, stream);
    ksr_set_error("pointwise_bwd: unsupported n_out=%d", n_out);
    return KSR_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// The same data gradients with the weight / bias gradient of the 1x1 conv accumulated in the same pass (bf16 storage mode):
// both operands of that product are already in the kernel's registers, so the separate ksr_wgrad launch and its
// second read of them go away.  slab: ksr_fused_wgrad_slab_floats() floats of workspace.
extern "C" int64_t ksr_fused_wgrad_slab_floats(void) { return (int64_t)KSR_WGF_MAX_GRID * KSR_WGF_ROW; }

extern "C" int ksr_gate_pw_bwd_wg(const float* dz, const float* fn2_w, const float* gate_saved, const float* t_saved,
                                  float* dt_value, float* dgate, float* slab, float* dW, float* db, int B, int H, int W,
                                  hipStream_t stream) {
    KSR_CHECK_ARG(dz && fn2_w && gate_saved && t_saved && dt_value && dgate, "gate_pw_bwd_wg: null pointer");
    if (int rc = dims_ok(B, H, W, "gate_pw_bwd_wg")) return rc;
    BwdArgs a{};
    a.dz = dz; a.dz_ld = 64; a.wt = fn2_w; a.e1 = gate_saved; a.e1_ld = 64; a.e2 = t_saved; a.e2_ld = 128;
    a.out1 = dt_value; a.out1_ld = 128; a.out1_c0 = 0; a.out2 = dgate; a.wg_slab = slab; a.B = B; a.H = H; a.W = W;
    return launch_dgrad_wgf<BPRO_PLAIN, BEPI_GATE>(a, dW, db, stream);
}

extern "C" int ksr_pw2_bwd_wg(const float* dy, const float* pw2_w, const float* operand_saved, float* ds, float* slab,
                              float* dW, float* db, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dy && pw2_w && operand_saved && ds, "pw2_bwd_wg: null pointer");
    if (int rc = dims_ok(B, H, W, "pw2_bwd_wg")) return rc;
    BwdArgs a{};
    a.dz = dy; a.dz_ld = 64; a.wt = pw2_w; a.e1 = operand_saved; a.e1_ld = 64; a.out1 = ds; a.out1_ld = 64;
    a.wg_slab = slab; a.B = B; a.H = H; a.W = W;
    return launch_dgrad_wgf<BPRO_PLAIN, BEPI_RELUMASK>(a, dW, db, stream);
}

// ---------------------------------------------------------------------------------------------
// First gated block (ffn2, whose input is ffn1): the saved tensor t [N,128] = relu(fn_1(ffn1(Y))) is rebuilt from the
// per-pixel scalar v and the coefficients ab = A [128] | B [128] that ksr_stem_gate_fwd wrote, t[n][c] = relu(A[c] v[n] +
// B[c]), instead of being read (and, in the forward, written): the `_v` variants of the three backward entry points
// that consume it.
extern "C" int ksr_gate_pw_bwd_v(const float* dz, const float* fn2_w, const float* gate_saved, const float* v_saved,
                                 const float* ab, float* dt_value, float* dgate, float* slab, float* dW, float* db, int B,
                                 int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dz && fn2_w && gate_saved && v_saved && ab && dt_value && dgate, "gate_pw_bwd_v: null pointer");
    if (int rc = dims_ok(B, H, W, "gate_pw_bwd_v")) return rc;
    BwdArgs a{};
    a.dz = dz; a.dz_ld = 64; a.wt = fn2_w; a.e1 = gate_saved; a.e1_ld = 64; a.aff_v = v_saved; a.aff_ab = ab;
    a.out1 = dt_value; a.out1_ld = 128; a.out1_c0 = 0; a.out2 = dgate; a.wg_slab = slab; a.B = B; a.H = H; a.W = W;
    if (slab) return launch_dgrad_wgf<BPRO_PLAIN, BEPI_GATE, true>(a, dW, db, stream);     // bf16 storage: + fn_2's weight gradient
    return launch_dgrad<64, BPRO_PLAIN, BEPI_GATE, true>(a, stream);
}

extern "C" int ksr_fn1_bwd_stem_v(float* dt, const float* dgate, const float* gate_w, const float* v_saved, const float* ab,
                                  const float* fn1_w, const float* pw2_w, float* dpre, float* sums_slab, float* dt_sums,
                                  float* dx_sums, int dx_row_stride, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dt && dgate && gate_w && v_saved && ab && fn1_w && pw2_w && dpre && sums_slab && dt_sums && dx_sums &&
                  dx_row_stride >= 64, "fn1_bwd_stem_v: bad argument");
    if (int rc = dims_ok(B, H, W, "fn1_bwd_stem_v")) return rc;
    const long ntiles = (long)B * ((H + 3) / 4) * ((W + 31) / 32);
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);
    if (ksr_prec() == KSR_BF16S) {
        // bf16 storage: no matrix product, column sums on the matrix cores from bf16 tiles
        StemBwdArgs a{dt, dgate, gate_w, v_saved, ab, fn1_w, pw2_w, dpre, sums_slab, B, H, W};
        hipLaunchKernelGGL(fn1_stem_bwd_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, a);
        KSR_LAUNCH_CHECK("fn1_bwd_stem_v");
    } else {
        BwdArgs a{};
        a.dz = dt; a.dz_ld = 128; a.halo_src = dgate; a.dw_w = gate_w; a.aff_v = v_saved; a.aff_ab = ab;
        a.wt = fn1_w; a.B = B; a.H = H; a.W = W; a.v = v_saved; a.p2w = pw2_w; a.dpre = dpre; a.sums_slab = sums_slab;
        const int rc = ksr_prec() == KSR_BF16 ? launch_dgrad_p<KSR_BF16, 128, BPRO_CONVT_G, BEPI_STEM, true>(a, stream)
                                              : launch_dgrad_p<KSR_F32, 128, BPRO_CONVT_G, BEPI_STEM, true>(a, stream);
        if (rc) return rc;
    }
    ColRedArgs r{sums_slab, (int)grid, {dt_sums, dt_sums + 64, dx_sums}, {128, 128, dx_row_stride}};
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(3, 4), dim3(1024), 0, stream, r);
    KSR_LAUNCH_CHECK("fn1_bwd_stem_v(column sums)");
    return KSR_OK;
}
