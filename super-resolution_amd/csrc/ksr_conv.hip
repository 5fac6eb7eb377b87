// Separable-conv trunk of the hyper-encoder, forward: every kernel is
//     [prologue that builds a 64-channel operand per pixel]  ->  1x1-conv GEMM on MFMA  ->  activation
// on NHWC activations, one pixel per lane (see ksr_gemm.h).
//
//   PRO_PLAIN  : operand = in[n, 0:64]
//   PRO_STEM   : operand = pointwise2(relu(dw3x3(pointwise1(Y))))  - the whole first FFN (1 -> 1 -> 64),
//                read straight from the Y plane of the NCHW input image
//   PRO_DWRELU : operand = relu(dw3x3(in) + b)                     - FFN middle (depthwise + ReLU)
//   PRO_GATE   : operand = in[n, 0:64] * (dw3x3(in[:, 64:128]) + b) - GatedFFN value * gated half
// The depthwise 3x3 (zero padding 1) reads a 6 x 34 pixel halo tile of its 64 input channels staged in
// LDS with coalesced 16-byte loads; the workgroup's 4 waves each take one 32-pixel row of a 4 x 32 tile.
#include "ksr_conv_common.h"

namespace {

enum { PRO_PLAIN = 0, PRO_STEM = 1, PRO_DWRELU = 2, PRO_GATE = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

struct ConvArgs {
    const float* in;      // NHWC [N, in_ld] (PLAIN/DWRELU/GATE) or NCHW image (STEM, plane 0 is Y)
    long img_bstride;     // STEM: elements between images
    int in_ld;            // channels per pixel of `in`
    const float* wt;      // [NOUT, 64]
    const float* bias;    // [NOUT]
    const float* dw_w;    // [64, 9]   (DWRELU/GATE)  |  STEM: [9]
    const float* dw_b;    // [64]                      |  STEM: [1]
    const float* p1;      // STEM: pointwise1 {w, b} as two pointers
    const float* p1b;
    const float* p2w;     // STEM: pointwise2 weight [64], bias [64]
    const float* p2b;
    float* out;           // NHWC [N, NOUT]
    float* save;          // training: STEM v [N]; DWRELU operand [N,64]; GATE gate G [N,64]; else null
    float* out32;         // optional fp32 copy of the output (whatever the storage mode), else null
    int B, H, W;
};


template <int PREC, int PRO, int NB>
struct __attribute__((aligned(16))) ConvSmem {
    float frag[ksr_frag_floats_p<PREC>(32, NB)];
    float biasfrag[NB * 32];
    // (split fp32, stencil kernels: the depthwise table lives in the halo tile's padding, ksr_stage_dw_pad)
    float dww[((PRO == PRO_DWRELU || PRO == PRO_GATE) && PREC != KSR_F32X3) ? 9 * 64 : 16];
    float dwb[64];
    float p2w[PRO == PRO_STEM ? 64 : 4];
    float p2b[PRO == PRO_STEM ? 64 : 4];
    float halo[(PRO == PRO_DWRELU || PRO == PRO_GATE) ? HALO_H * HALO_W * HALO_LD : 4];
    // one LDS transposer per wave for the coalesced loads / stores; the stencil kernels reuse the halo tile
    // for it once every wave has finished reading the tile
    float xpose[(PRO == PRO_DWRELU || PRO == PRO_GATE) ? 4 : 4 * ksr_xpose_floats_p<PREC>()];
};
static_assert(4 * KSR_XPOSE_FLOATS <= HALO_H * HALO_W * HALO_LD, "transposers must fit in the halo tile");

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_SIGMOID) return ksr_sigmoid(v);
    return v;
}

template <int PREC, int PRO, int NOUT, int ACT>
__global__ void __launch_bounds__(256, 2) conv_fwd_kernel(ConvArgs a) {
    constexpr int NB = NOUT / 32;   // 16-register blocks per lane half (NOUT/2 outputs per half)
    using ST = typename KsrStore<PREC>::T;
    __shared__ ConvSmem<PREC, PRO, NB> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;

    ksr_stage_weights<PREC, 32, NB, false>(sm.frag, sm.biasfrag, a.wt, 64, a.bias, MapHL{NOUT / 2}, MapHL{32}, tid, 256);
    if (PRO == PRO_DWRELU || PRO == PRO_GATE) {
        if constexpr (PREC == KSR_F32X3) ksr_stage_dw_pad<false>(sm.halo, a.dw_w, tid);
        else ksr_stage_dw<false>(sm.dww, a.dw_w, tid);
        if (tid < 64) sm.dwb[tid] = a.dw_b[tid];
    }
    if (PRO == PRO_STEM) {
        if (tid < 64) { sm.p2w[tid] = a.p2w[tid]; sm.p2b[tid] = a.p2b[tid]; }
        if (tid < 9) sm.dww[tid] = a.dw_w[tid];
    }
    __syncthreads();

    const int HW = a.H * a.W;
    const long N = (long)a.B * HW;
    constexpr bool spatial = (PRO == PRO_DWRELU || PRO == PRO_GATE);
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = spatial ? (long)a.B * tiles_y * tiles_x : (N + 127) / 128;

    // plain 1x1 conv (bf16 storage, or fp32 up to 128 outputs): the next tile's rows are requested before this tile's product
    constexpr bool AHEAD = (PRO == PRO_PLAIN) && (sizeof(ST) == 2 || NOUT <= 128);
    KsrTile<ST> ahead;
    auto fetch_tile = [&](long tile) {
        const long n0 = tile * 128 + wv * 32;
        const int nv = (N - n0) < 32 ? (int)(N - n0 > 0 ? N - n0 : 0) : 32;
        ksr_co_fetch<ST, true>(a.in, n0, nv, a.in_ld, 0, lane, ahead);
    };
    // stencil kernels walk the tiles so that neighbours share an L2; the 1x1 kernels keep the plain grid-stride walk
    const KsrTileWalkT<spatial> tw(ntiles);
    if (AHEAD && tw.first() < tw.end()) fetch_tile(tw.first());
    // stencil kernels with bf16 storage: likewise for the halo tile (7 x 16 bytes per thread)
    constexpr bool HAHEAD = spatial && sizeof(ST) == 2;
    KsrHalo<ST> hr;
    auto fetch_halo = [&](long t) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const int tb = (int)(t / ((long)tiles_x * tiles_y));
        ksr_halo_fetch<ST>(a.in, a.in_ld, (PRO == PRO_GATE) ? 64 : 0, tb, ty * 4, tx * 32, a.H, a.W, tid, hr);
    };
    if (HAHEAD && tw.first() < tw.end()) fetch_halo(tw.first());

    for (long tile = tw.first(); tile < tw.end(); tile += tw.step()) {
        int b, y, x;
        bool live;
        long n;
        if (spatial) {
            const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
            b = (int)(tile / ((long)tiles_x * tiles_y));
            const int y0 = ty * 4, x0 = tx * 32;
            y = y0 + wv; x = x0 + tok;
            live = (y < a.H) && (x < a.W);
            n = (long)b * HW + (long)y * a.W + x;
            // ---- stage the halo tile (rows y0-1..y0+4, cols x0-1..x0+32, 64 channels) ----
            __syncthreads();   // previous tile's readers are done
            if (!HAHEAD) fetch_halo(tile);
            ksr_halo_commit<ST>(sm.halo, tid, hr);
            __syncthreads();
            if (HAHEAD && tile + tw.step() < tw.end()) fetch_halo(tile + tw.step());     // in flight during this tile
        } else {
            n = tile * 128 + wv * 32 + tok;
            live = n < N;
            const long nn = live ? n : 0;
            b = (int)(nn / HW);
            const int hw = (int)(nn % HW);
            y = hw / a.W; x = hw % a.W;
        }

        static_assert(PREC != KSR_F32X3 || spatial || (PRO == PRO_PLAIN && AHEAD), "split fp32: stencil kernels, or the plain 1x1 conv with prefetched tiles");
        // split fp32: the plain kernel's transposers are half width; the stencil kernels' are the (full-size) halo tile
        constexpr int OUTP = spatial ? (int)KSR_F32 : PREC;
        float* xb = spatial ? sm.halo + wv * KSR_XPOSE_FLOATS : sm.xpose + wv * ksr_xpose_floats_p<PREC>();
        // first row and number of real rows of this wave's 32-pixel run (consecutive pixel indices)
        const long n0w = spatial ? (long)b * HW + (long)y * a.W + (x - tok) : tile * 128 + wv * 32;
        int nvalid;
        if (spatial) nvalid = (y < a.H) ? (a.W - (x - tok) < 32 ? a.W - (x - tok) : 32) : 0;
        else nvalid = (N - n0w) < 32 ? (int)(N - n0w > 0 ? N - n0w : 0) : 32;
        float op[32];
        if (PRO == PRO_PLAIN) {
            if (AHEAD) {
                if constexpr (PREC == KSR_F32X3) ksr_co_finish_half(xb, lane, ahead, op);
                else ksr_co_finish<ST>(xb, lane, ahead, op);
                if (tile + tw.step() < tw.end()) fetch_tile(tile + tw.step());
            } else {
                ksr_load32_co<ST>(xb, a.in, n0w, nvalid, a.in_ld, 0, lane, op);
            }
        } else if (PRO == PRO_STEM) {
            const float w1 = a.p1[0], b1 = a.p1b[0];
            const float* Y = a.in + (long)b * a.img_bstride;
            float v = a.dw_b[0];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    float u = 0.f;
                    if (live && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) u = w1 * Y[(long)yy * a.W + xx] + b1;
                    v += sm.dww[(dy + 1) * 3 + (dx + 1)] * u;
                }
            v = v > 0.f ? v : 0.f;
            if (a.save && live && h == 0) a.save[n] = v;
#pragma unroll
            for (int i = 0; i < 32; ++i) op[i] = sm.p2w[32 * h + i] * v + sm.p2b[32 * h + i];
        } else {
            // depthwise 3x3 over the halo tile, this lane's 32 channels
            float g[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) g[i] = sm.dwb[32 * h + i];
            if constexpr (PREC == KSR_F32X3) ksr_dwconv_lane_pad(sm.halo, wv, tok, h, g);
            else ksr_dwconv_lane(sm.halo, sm.dww, wv, tok, h, g);
            __syncthreads();       // every wave is done with the halo tile: it now serves as the transposers
            if (PRO == PRO_DWRELU) {
#pragma unroll
                for (int i = 0; i < 32; ++i) op[i] = g[i] > 0.f ? g[i] : 0.f;
                if (a.save) ksr_store32_co<ST, true, true>(xb, a.save, n0w, nvalid, 64, 0, lane, op);
            } else {
                if (a.save) ksr_store32_co<ST, true, true>(xb, a.save, n0w, nvalid, 64, 0, lane, g);
                float val[32];
                ksr_load32_co<ST>(xb, a.in, n0w, nvalid, a.in_ld, 0, lane, val);
#pragma unroll
                for (int i = 0; i < 32; ++i) op[i] = val[i] * g[i];
            }
        }

        f32x16 acc[NB];
        ksr_wave_gemm<PREC, 32, NB>(sm.frag, sm.biasfrag, op, acc, lane);

#pragma unroll
        for (int grp = 0; grp < NB / 2; ++grp) {
            float o[32];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                o[q] = apply_act<ACT>(acc[2 * grp][q]);
                o[16 + q] = apply_act<ACT>(acc[2 * grp + 1][q]);
            }
            ksr_store32_co_p<OUTP, ST, !AHEAD>(xb, a.out, n0w, nvalid, NOUT, 64 * grp, lane, o);
            if (a.out32) ksr_store32_co_p<OUTP, float, !AHEAD, true>(xb, a.out32, n0w, nvalid, NOUT, 64 * grp, lane, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ffn1 + ffn2 + ffn3.pointwise1 in one kernel, reading only the Y plane.
// ffn1's output is x1[k] = p2w[k] v + p2b[k] with ONE scalar v = relu(dw3x3(pw1(Y))) per pixel, so ffn2.fn_1 applied to
// it is affine in v per channel: t[c] = relu(A[c] v + B[c]), A = fn1_w p2w, B = fn1_w p2b + fn1_b (formed once per
// workgroup).  The 128-channel tensor t therefore needs no GEMM and no HBM round trip: the gate's depthwise 3x3 is
// evaluated from the 6 x 34 window of v, which comes from an 8 x 36 window of Y.  Then value * gate -> fn_2 (MFMA) ->
// sigmoid -> ffn3.pointwise1 (MFMA), the second product taking the first one's result straight from registers.
struct StemGateArgs {
    const float* img; long img_bstride;                       // NCHW image, plane 0 = Y
    const float *p1w, *p1b, *dw1w, *dw1b, *p2w, *p2b;          // ffn1: pointwise1 {w,b}, depthwise [9], [1], pointwise2 [64],[64]
    const float *fn1w, *fn1b;                                  // ffn2.fn_1 [128,64], [128]
    const float *gw, *gb;                                      // ffn2.gate [64,9], [64]
    const float *fn2w, *fn2b;                                  // ffn2.fn_2 [64,64], [64]
    const float *pw1w, *pw1b;                                  // ffn3.pointwise1 [64,64], [64]
    float* t3;                                                 // [N,64] ffn3.pointwise1 output
    float *sv_v, *sv_t, *sv_g, *sv_x3;                         // training saves: v [N], t [N,128] (optional), gate [N,64], x3 [N,64]
    float* sv_ab;                                              // training, optional: A [128] | B [128] of t = relu(A v + B)
    int B, H, W;
};

constexpr int SG_YH = 8, SG_YW = 36, SG_VH = 6, SG_VW = 34;

template <int PREC>
struct __attribute__((aligned(16))) StemGateSmem {
    float frag1[ksr_frag_floats_p<PREC>(32, 2)];
    float frag2[ksr_frag_floats_p<PREC>(32, 2)];
    float bias1[64], bias2[64];
    float dww[9 * 64], dwb[64];
    float A[128], Bc[128];
    float u[SG_YH * SG_YW];            // pw1(Y) inside the image, 0 outside
    float v[SG_VH * SG_VW];            // relu(dw3x3(u)) inside the image, -1 outside (v >= 0 inside)
    float xpose[4 * ksr_xpose_floats_p<PREC>()];
};

template <int PREC>
__global__ void __launch_bounds__(256, 2) stem_gate_fwd_kernel(StemGateArgs a) {
    using ST = typename KsrStore<PREC>::T;
    __shared__ StemGateSmem<PREC> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;

    ksr_stage_weights<PREC, 32, 2, false>(sm.frag1, sm.bias1, a.fn2w, 64, a.fn2b, MapHL{32}, MapHL{32}, tid, 256);
    ksr_stage_weights<PREC, 32, 2, false>(sm.frag2, sm.bias2, a.pw1w, 64, a.pw1b, MapHL{32}, MapHL{32}, tid, 256);
    ksr_stage_dw<false>(sm.dww, a.gw, tid);
    if (tid < 64) sm.dwb[tid] = a.gb[tid];
    if (tid < 128) {
        // row tid of fn_1's weight as 16 independent 16-byte loads (one memory latency, not 64)
        float4 wr[16], pw[16], pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            wr[q] = reinterpret_cast<const float4*>(a.fn1w + tid * 64)[q];
            pw[q] = reinterpret_cast<const float4*>(a.p2w)[q];
            pb[q] = reinterpret_cast<const float4*>(a.p2b)[q];
        }
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            sa = fmaf(wr[q].x, pw[q].x, sa); sb = fmaf(wr[q].x, pb[q].x, sb);
            sa = fmaf(wr[q].y, pw[q].y, sa); sb = fmaf(wr[q].y, pb[q].y, sb);
            sa = fmaf(wr[q].z, pw[q].z, sa); sb = fmaf(wr[q].z, pb[q].z, sb);
            sa = fmaf(wr[q].w, pw[q].w, sa); sb = fmaf(wr[q].w, pb[q].w, sb);
        }
        sm.A[tid] = sa;
        sm.Bc[tid] = sb + a.fn1b[tid];
        if (a.sv_ab && blockIdx.x == 0) { a.sv_ab[tid] = sa; a.sv_ab[128 + tid] = sb + a.fn1b[tid]; }
    }
    const float w1 = a.p1w[0], b1 = a.p1b[0], db1 = a.dw1b[0];
    float k1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k1[t] = a.dw1w[t];
    __syncthreads();

    const int HW = a.H * a.W;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    float* xb = sm.xpose + wv * ksr_xpose_floats_p<PREC>();

    // the Y window of a tile (8 x 36 values, at most two per thread) is requested one tile ahead: the only global
    // reads of the kernel would otherwise expose their full latency once per tile
    float ynext[2];
    auto fetch_y = [&](long t) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const int b = (int)(t / ((long)tiles_x * tiles_y));
        const float* Y = a.img + (long)b * a.img_bstride;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            const int yy = ty * 4 - 2 + i / SG_YW, xx = tx * 32 - 2 + i % SG_YW;
            const bool in = i < SG_YH * SG_YW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            ynext[k] = in ? w1 * Y[(long)yy * a.W + xx] + b1 : 0.f;
        }
    };
    const KsrTileWalk tw = ksr_xcd_tile_walk(ntiles);
    if (tw.first < tw.end) fetch_y(tw.first);

    for (long tile = tw.first; tile < tw.end; tile += tw.step) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const int b = (int)(tile / ((long)tiles_x * tiles_y));
        const int y0 = ty * 4, x0 = tx * 32;
        const int y = y0 + wv, x = x0 + tok;
        const bool live = (y < a.H) && (x < a.W);
        const long n = (long)b * HW + (long)y * a.W + x;
        __syncthreads();                       // the previous tile's readers of u / v are done
        sm.u[tid] = ynext[0];
        if (tid + 256 < SG_YH * SG_YW) sm.u[tid + 256] = ynext[1];
        if (tile + tw.step < tw.end) fetch_y(tile + tw.step);     // in flight during this tile
        __syncthreads();
        if (tid < SG_VH * SG_VW) {
            const int vy = tid / SG_VW, vx = tid % SG_VW;
            const int yy = y0 - 1 + vy, xx = x0 - 1 + vx;
            float v = db1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v += k1[dy * 3 + dx] * sm.u[(vy + dy) * SG_YW + vx + dx];
            v = v > 0.f ? v : 0.f;
            sm.v[tid] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? v : -1.f;
        }
        __syncthreads();

        // gate = dw3x3(relu(A v + B)) over this lane's 32 gate channels (64 + 32h + i of t)
        float g[32], tg[32];
        {
            float Ag[32], Bg[32];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&sm.A[64 + 32 * h + 4 * q]);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&sm.Bc[64 + 32 * h + 4 * q]);
                const f32x4 gb = *reinterpret_cast<const f32x4*>(&sm.dwb[32 * h + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { Ag[4 * q + e] = av[e]; Bg[4 * q + e] = bv[e]; g[4 * q + e] = gb[e]; }
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float vn = sm.v[(wv + dy) * SG_VW + tok + dx];
                    const float* wp = &sm.dww[(dy * 3 + dx) * 64 + 32 * h];
                    if (dy == 1 && dx == 1) {
#pragma unroll
                        for (int i = 0; i < 32; ++i) tg[i] = 0.f;
                    }
                    // a neighbour outside the image contributes nothing (t is zero-padded): one branch per tap, taken by
                    // the lanes whose neighbour exists, instead of a select per channel
                    if (vn >= 0.f) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const f32x4 wq = *reinterpret_cast<const f32x4*>(wp + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = fmaxf(fmaf(Ag[4 * q + e], vn, Bg[4 * q + e]), 0.f);
                                if (dy == 1 && dx == 1) tg[4 * q + e] = t;
                                g[4 * q + e] = fmaf(wq[e], t, g[4 * q + e]);
                            }
                        }
                    }
                }
        }
        const float vc = sm.v[(wv + 1) * SG_VW + tok + 1];     // this pixel's v (>= 0 when live)
        float op[32], val[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&sm.A[32 * h + 4 * q]);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&sm.Bc[32 * h + 4 * q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = fmaf(av[e], vc, bv[e]);
                val[4 * q + e] = t > 0.f ? t : 0.f;
                op[4 * q + e] = val[4 * q + e] * g[4 * q + e];
            }
        }
        const long n0w = (long)b * HW + (long)y * a.W + x0;
        const int nvalid = (y < a.H) ? (a.W - x0 < 32 ? a.W - x0 : 32) : 0;
        if (a.sv_v) {
            if (live && h == 0) a.sv_v[n] = vc;
            if (a.sv_t) {
                ksr_store32_co_p<PREC, ST, true, true>(xb, a.sv_t, n0w, nvalid, 128, 0, lane, val);
                ksr_store32_co_p<PREC, ST, true, true>(xb, a.sv_t, n0w, nvalid, 128, 64, lane, tg);
            }
            ksr_store32_co_p<PREC, ST, true, true>(xb, a.sv_g, n0w, nvalid, 64, 0, lane, g);
        }
        f32x16 acc[2];
        ksr_wave_gemm<PREC, 32, 2>(sm.frag1, sm.bias1, op, acc, lane);
        float x3[32];
#pragma unroll
        for (int q = 0; q < 16; ++q) { x3[q] = ksr_sigmoid(acc[0][q]); x3[16 + q] = ksr_sigmoid(acc[1][q]); }
        if (a.sv_x3) {
            ksr_store32_co_p<PREC, ST, true, true>(xb, a.sv_x3, n0w, nvalid, 64, 0, lane, x3);
            if (sizeof(ST) == 2) {
                // the backward and the next layer read x3 as stored: continue from the rounded values
#pragma unroll
                for (int i = 0; i < 32; ++i) x3[i] = (float)(__bf16)x3[i];
            }
        }
        ksr_wave_gemm<PREC, 32, 2>(sm.frag2, sm.bias2, x3, acc, lane);
        float o[32];
#pragma unroll
        for (int q = 0; q < 16; ++q) { o[q] = acc[0][q]; o[16 + q] = acc[1][q]; }
        ksr_store32_co_p<PREC, ST>(xb, a.t3, n0w, nvalid, 64, 0, lane, o);
    }
}

template <int PREC>
int launch_stem_gate_p(const StemGateArgs& a, hipStream_t s) {
    const long ntiles = (long)a.B * ((a.H + 3) / 4) * ((a.W + 31) / 32);
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);
    hipLaunchKernelGGL((stem_gate_fwd_kernel<PREC>), dim3(grid), dim3(256), 0, s, a);
    KSR_LAUNCH_CHECK("stem_gate_fwd");
    return KSR_OK;
}

template <int PREC, int PRO, int NOUT, int ACT>
int launch_conv_p(const ConvArgs& a, hipStream_t s) {
    const bool spatial = (PRO == PRO_DWRELU || PRO == PRO_GATE);
    const long N = (long)a.B * a.H * a.W;
    const long ntiles = spatial ? (long)a.B * ((a.H + 3) / 4) * ((a.W + 31) / 32) : (N + 127) / 128;
    // stencil kernels: two workgroups per CU (halo tiles); plain 1x1 kernels: one round of persistent workgroups
    const unsigned grid = spatial ? (unsigned)(ntiles < 512 ? ntiles : 512)
                                  : ksr_persistent_grid(conv_fwd_kernel<PREC, PRO, NOUT, ACT>, ntiles, 1024);
    hipLaunchKernelGGL((conv_fwd_kernel<PREC, PRO, NOUT, ACT>), dim3(grid), dim3(256), 0, s, a);
    KSR_LAUNCH_CHECK("conv_fwd");
    return KSR_OK;
}

template <int PRO, int NOUT, int ACT>
int launch_conv(const ConvArgs& a, hipStream_t s) {
#define KSR_CALL(P) return launch_conv_p<P, PRO, NOUT, ACT>(a, s)
    if constexpr ((PRO == PRO_PLAIN && NOUT <= 128) || ((PRO == PRO_DWRELU || PRO == PRO_GATE) && NOUT == 64)) {      // (192 outputs: the three weight planes would leave one workgroup per CU)
        if (ksr_prec() == KSR_F32 && ksr_f32_split()) KSR_CALL(KSR_F32X3);      // fp32 products on the bf16 matrix cores
    }
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    return KSR_OK;
}

int check_dims(int B, int H, int W, const char* who) {
    KSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "%s: non-positive dimension", who);
    KSR_CHECK_ARG((long)B * H * W < (1L << 31), "%s: more than 2^31 pixels", who);
    return KSR_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int ksr_stem_fwd(const float* x_img, int64_t img_batch_stride, const float* pw1_w, const float* pw1_b,
                            const float* dw_w, const float* dw_b, const float* pw2_w, const float* pw2_b,
                            const float* fn1_w, const float* fn1_b, float* t_out, float* save_v, int B, int H,
                            int W, hipStream_t stream) {
    KSR_CHECK_ARG(x_img && pw1_w && pw1_b && dw_w && dw_b && pw2_w && pw2_b && fn1_w && fn1_b && t_out, "stem_fwd: null pointer");
    if (int rc = check_dims(B, H, W, "stem_fwd")) return rc;
    ConvArgs a{};
    a.in = x_img; a.img_bstride = img_batch_stride; a.wt = fn1_w; a.bias = fn1_b;
    a.dw_w = dw_w; a.dw_b = dw_b; a.p1 = pw1_w; a.p1b = pw1_b; a.p2w = pw2_w; a.p2b = pw2_b;
    a.out = t_out; a.save = save_v; a.B = B; a.H = H; a.W = W;
    return launch_conv<PRO_STEM, 128, ACT_RELU>(a, stream);
}

extern "C" int ksr_pointwise_fwd(const float* in, const float* w, const float* bias, float* out, int n_out,
                                 int act, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(in && w && bias && out, "pointwise_fwd: null pointer");
    if (int rc = check_dims(B, H, W, "pointwise_fwd")) return rc;
    ConvArgs a{};
    a.in = in; a.in_ld = 64; a.wt = w; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W;
    if (n_out == 64 && act == ACT_NONE) return launch_conv<PRO_PLAIN, 64, ACT_NONE>(a, stream);
    if (n_out == 128 && act == ACT_RELU) return launch_conv<PRO_PLAIN, 128, ACT_RELU>(a, stream);
    if (n_out == 192 && act == ACT_NONE) return launch_conv<PRO_PLAIN, 192, ACT_NONE>(a, stream);
    ksr_set_error("pointwise_fwd: unsupported (n_out=%d, act=%d)", n_out, act);
    return KSR_ERR_UNSUPPORTED;
}

extern "C" int ksr_dwrelu_pw_fwd(const float* in, const float* dw_w, const float* dw_b, const float* w,
                                 const float* bias, float* out, float* save_operand, int B, int H, int W,
                                 hipStream_t stream) {
    KSR_CHECK_ARG(in && dw_w && dw_b && w && bias && out, "dwrelu_pw_fwd: null pointer");
    if (int rc = check_dims(B, H, W, "dwrelu_pw_fwd")) return rc;
    ConvArgs a{};
    a.in = in; a.in_ld = 64; a.dw_w = dw_w; a.dw_b = dw_b; a.wt = w; a.bias = bias; a.out = out;
    a.save = save_operand; a.B = B; a.H = H; a.W = W;
    return launch_conv<PRO_DWRELU, 64, ACT_NONE>(a, stream);
}

extern "C" int ksr_gate_pw_fwd(const float* t_in, const float* gate_w, const float* gate_b, const float* w,
                               const float* bias, float* out, float* save_gate, float* out_f32_copy, int B, int H, int W,
                               hipStream_t stream) {
    KSR_CHECK_ARG(t_in && gate_w && gate_b && w && bias && out, "gate_pw_fwd: null pointer");
    if (int rc = check_dims(B, H, W, "gate_pw_fwd")) return rc;
    ConvArgs a{};
    a.in = t_in; a.in_ld = 128; a.dw_w = gate_w; a.dw_b = gate_b; a.wt = w; a.bias = bias; a.out = out;
    a.save = save_gate; a.out32 = out_f32_copy; a.B = B; a.H = H; a.W = W;
    return launch_conv<PRO_GATE, 64, ACT_SIGMOID>(a, stream);
}

extern "C" int ksr_stem_gate_fwd(const float* x_img, int64_t img_batch_stride, const float* const* params, float* t3_out,
                                 float* save_v, float* save_t, float* save_gate, float* save_x3, float* save_ab, int B, int H,
                                 int W, hipStream_t stream) {
    KSR_CHECK_ARG(x_img && params && t3_out, "stem_gate_fwd: null pointer");
    for (int i = 0; i < 14; ++i) KSR_CHECK_ARG(params[i] != nullptr, "stem_gate_fwd: params[%d] is null", i);
    const bool any = save_v || save_t || save_gate || save_x3 || save_ab, all = save_v && save_gate && save_x3 && (save_t || save_ab);
    KSR_CHECK_ARG(!any || all, "stem_gate_fwd: training passes save_v, save_gate, save_x3 and save_t and / or save_ab; inference none");
    if (int rc = check_dims(B, H, W, "stem_gate_fwd")) return rc;
    StemGateArgs a{};
    a.img = x_img; a.img_bstride = img_batch_stride;
    a.p1w = params[0]; a.p1b = params[1]; a.dw1w = params[2]; a.dw1b = params[3]; a.p2w = params[4]; a.p2b = params[5];
    a.fn1w = params[6]; a.fn1b = params[7]; a.gw = params[8]; a.gb = params[9]; a.fn2w = params[10]; a.fn2b = params[11];
    a.pw1w = params[12]; a.pw1b = params[13];
    a.t3 = t3_out; a.sv_v = save_v; a.sv_t = save_t; a.sv_g = save_gate; a.sv_x3 = save_x3; a.sv_ab = save_ab;
    a.B = B; a.H = H; a.W = W;
#define KSR_CALL(P) return launch_stem_gate_p<P>(a, stream)
    if (ksr_prec() == KSR_F32 && ksr_f32_split()) KSR_CALL(KSR_F32X3);      // fp32 products on the bf16 matrix cores
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    return KSR_OK;
}
