// feed (1x1 conv 64 -> P) + per-pixel KAN evaluation + pixel shuffle, fused: the [N, P] tensor of
// per-pixel spline coefficients lives only in MFMA accumulators.
//
// The P = O*(Kc+2) output channels of `feed` are assigned to the two lanes of a pixel so that lane
// half h receives, for each of ITS O/2 outputs o, the complete parameter set (coef[o,0..Kc-1],
// uw[o], rw[o]) (MapKan in ksr_gemm.h).  The KAN layer is then lane-local:
//   y[o] = rw[o]*silu(x) + uw[o] * sum_g B_g(x) coef[o,g]
// and with O = r*r the lane's outputs are r/2 rows of r contiguous pixels of the x r image, written
// with one 8/16-byte store per row directly into the Y plane of the [B,3,rH,rW] result.
#include "ksr_gemm.h"

namespace {

struct FeedArgs {
    const float* feat;     // [N, 64]
    const float* x_img;    // NCHW image, plane 0 = Y
    long img_bstride;
    const float* w;        // [P, 64]
    const float* bias;     // [P]
    float* out;            // [B, C_out, rH, rW]; plane 0 is written
    long out_bstride;
    float* save_us;        // training: [N, O, 2] = (uw, spline sum) per output, may be null
    // backward
    const float* gout;     // gradient of out (same layout/strides as out)
    float* gfeat;          // [N, 64]
    float* gparam;         // [N, 2, NB*16] gradient of the feed output in lane order (half, e)
    int B, H, W;
    // Per-image weights (the channel-attention head, whose 1x1 map differs from image to image): w / bias advance by
    // w_bstride / bias_bstride elements per image, blockIdx.y is the image and the tiles cover its HW pixels.
    // 0 / 0 with a one-dimensional grid: one weight matrix, tiles over all N pixels.
    long w_bstride, bias_bstride;
    int feat_ld, feat_c0;      // row length and first channel of feat (64, 0 for a dense [N,64] tensor)
    int gfeat_ld, gfeat_c0;    // the same for gfeat
    // backward, optional: feat is the output of a LayerNorm (the encoder's final norm) whose backward runs here too: gfeat
    // then receives the gradient of that LayerNorm's INPUT, and lnf_slab [grid][2][64] its gain / shift gradient sums
    const void* lnf_xhat; const float* lnf_rstd; const float* lnf_w; float* lnf_slab;
};

// Pixels a workgroup's tiles run over: all N, or the HW pixels of image blockIdx.y (per-image weights)
struct FeedRange { long first, count; };
__device__ __forceinline__ FeedRange feed_range(const FeedArgs& a) {
    const long HW = (long)a.H * a.W;
    if (a.w_bstride == 0 && gridDim.y == 1) return FeedRange{0, (long)a.B * HW};
    return FeedRange{(long)blockIdx.y * HW, HW};
}

template <int KC>
__device__ __forceinline__ void dense_basis(int m, const float (&Nv)[KSR_MAX_ORDER + 1], int K, float (&bv)[KC]) {
#pragma unroll
    for (int g = 0; g < KC; ++g) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q <= KSR_MAX_ORDER; ++q) v = (m >= 0 && q <= K && g == m - K + q) ? Nv[q] : v;
        bv[g] = v;
    }
}

template <int KORD>
__device__ __forceinline__ int basis_any(float x, const float* t, int M, float (&Nv)[KSR_MAX_ORDER + 1]) {
    float N[KORD + 1], dN[KORD + 1];
    const int m = ksr_bspline_local<KORD, false>(x, t, M, N, dN);
#pragma unroll
    for (int q = 0; q <= KSR_MAX_ORDER; ++q) Nv[q] = 0.f;
#pragma unroll
    for (int q = 0; q <= KORD; ++q) Nv[q] = N[q];
    return m;
}

template <int R, int KC>
struct FeedCfg {
    static constexpr int O = R * R, OH = O / 2, PER = KC + 2, NH = OH * PER;
    static constexpr int NB = (NH + 15) / 16;
};

template <int NB, bool COAL, int FRAG_FLOATS = NB * 32 * 64 / (COAL ? 2 : 1)>
struct __attribute__((aligned(16))) FeedSmem {
    float frag[FRAG_FLOATS];                          // bf16 storage implies bf16 fragments: half the bytes; split fp32: 1.5 x
    float biasfrag[NB * 32];
    float knots[KSR_MAX_KNOTS];
    float xpose[COAL ? 4 * 32 * KSR_XPOSE16_LD : 4];  // per-wave transposers of the coalesced (uw, S) stores
    // bf16 storage: per wave two staging buffers for the LDS-DMA of its 32 feature rows (128 bytes each, 16-byte chunk c of
    // row r at chunk position c ^ ((r >> 1) & 7): conflict-free row reads) and of the 32 pixels' luma values
    __bf16 stage[COAL ? 4 * 2 * 32 * 64 : 8];
    float xstage[COAL ? 4 * 2 * 64 : 4];
};

template <int PREC, int R, int KC, int KORD>
__global__ void __launch_bounds__(256, 2) feed_kan_fwd_kernel(FeedArgs a, KsrKnots kn) {
    using C = FeedCfg<R, KC>;
    constexpr int NB = C::NB;
    using ST = typename KsrStore<PREC>::T;
    // bf16 storage: the wave's 32 feature rows are fetched with lane-linear 16-byte loads one tile ahead of their use
    // (the kernel is bound by memory latency there).  fp32 storage (the exact mode is bound by the fp32 matrix pipe and
    // runs three waves per SIMD): every lane loads its own 128 bytes, which needs neither the LDS nor the registers.
    constexpr bool COAL = sizeof(ST) == 2;
    __shared__ FeedSmem<NB, COAL, (PREC == KSR_F32X3 ? ksr_frag_floats_p<KSR_F32X3>(32, NB) : NB * 32 * 64 / (COAL ? 2 : 1))> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, 32, NB, false>(sm.frag, sm.biasfrag, a.w + (long)blockIdx.y * a.w_bstride, 64,
                                           a.bias + (long)blockIdx.y * a.bias_bstride, MapKan{C::O, KC}, MapHL{32}, tid, 256);
    if (tid < KSR_MAX_KNOTS) sm.knots[tid] = kn.t[tid];
    __syncthreads();

    const int HW = a.H * a.W;
    const FeedRange rg = feed_range(a);
    const long N = rg.count;                         // pixels of this workgroup's range, which starts at rg.first
    const long ntiles = (N + 127) / 128;
    float* xb = sm.xpose + (COAL ? wv * 32 * KSR_XPOSE16_LD : 0);
    // bf16 storage: a tile's feature rows and luma values travel global -> LDS by LDS-DMA, requested one tile ahead.  The
    // transfers are issued from inline assembly and waited for explicitly, so the only vector-memory instructions the
    // compiler keeps count of are the tile's stores, for which nothing ever waits: a tile's stores drain while the next
    // tile is computed.  (With the tile fetched into registers the compiler's wait at the top of the loop was vmcnt(0):
    // loads and stores of a wave complete in issue order under ONE counter, so every tile waited for the previous tile's
    // stores - a full memory round trip per tile, 2.6 TB/s.)
    auto issue_tile = [&](long t, int buf) {
        const long r0 = t * 128 + wv * 32;
        const int nv = (N - r0) < 32 ? (int)(N - r0 > 0 ? N - r0 : 0) : 32;
        const long nb = rg.first + (nv > 0 ? r0 : 0);
        const int last = nv > 0 ? nv - 1 : 0;              // rows past the end re-read the last real row: never stored
        const int r8 = lane >> 3, cp = lane & 7;
        __bf16* st = sm.stage + (wv * 2 + buf) * 32 * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + r8, c = cp ^ ((r >> 1) & 7);
            const int rc = r < last ? r : last;
            ksr_dma16<true>(reinterpret_cast<const __bf16*>(a.feat) + (nb + rc) * a.feat_ld + a.feat_c0 + 8 * c, st + j * 512);
        }
        const long nt = nb + (tok < last ? tok : last);
        ksr_dma4(a.x_img + (nt / HW) * a.img_bstride + (nt % HW), sm.xstage + (wv * 2 + buf) * 64);
    };
    // One tile.  FULL (every tile but a ragged last one): all 128 pixels exist, nothing in the body is predicated and the
    // wave issues a fixed number of stores per tile (NST: 2 of the output, 4 of the (uw, S) pairs when they are kept).
    auto body = [&](long tile, int it, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const long nl = tile * 128 + wv * 32 + tok;
        const bool live = FULL || nl < N;
        const long n = rg.first + nl;
        const long nn = live ? n : rg.first;
        const int b = (int)(nn / HW), hw = (int)(nn % HW);
        const int y = hw / a.W, x = hw % a.W;

        float f[32];
        float xv;
        if constexpr (COAL) {
            // the transfers of this tile were issued before the previous tile's stores (none before the first tile)
            if (it == 0) ksr_dma_wait_but<0>();
            else if (a.save_us) ksr_dma_wait_but<(2 * C::OH == 16) ? R / 2 + 4 : 0>();
            else ksr_dma_wait_but<R / 2>();
            const __bf16* st = sm.stage + (wv * 2 + (it & 1)) * 32 * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ksr_unpack8(*reinterpret_cast<const uint4*>(st + tok * 64 + 8 * ((4 * h + q) ^ ((tok >> 1) & 7))), f + 8 * q);
            xv = sm.xstage[(wv * 2 + (it & 1)) * 64 + tok];
            if (!live) xv = 0.f;
            ksr_wave_fence();
            if (tile + gridDim.x < ntiles) issue_tile(tile + gridDim.x, (it + 1) & 1);
        } else {
            if (live) ksr_loadn<ST, 32>(a.feat, n * a.feat_ld + a.feat_c0 + 32 * h, f);
            else {
#pragma unroll
                for (int i = 0; i < 32; ++i) f[i] = 0.f;
            }
            xv = live ? a.x_img[(long)b * a.img_bstride + hw] : 0.f;
        }
        f32x16 acc[NB];
        ksr_wave_gemm<PREC, 32, NB>(sm.frag, sm.biasfrag, f, acc, lane);

        float Nv[KSR_MAX_ORDER + 1], bv[KC];
        const int m = basis_any<KORD>(xv, sm.knots, kn.M, Nv);
        dense_basis<KC>(m, Nv, KORD, bv);
        const float sl = ksr_silu(xv);

        float yv[C::OH], us[2 * C::OH];
#pragma unroll
        for (int ol = 0; ol < C::OH; ++ol) {
            float S = 0.f;
#pragma unroll
            for (int g = 0; g < KC; ++g) {
                const int e = ol * C::PER + g;
                S += bv[g] * acc[e / 16][e % 16];
            }
            const int eu = ol * C::PER + KC, er = eu + 1;
            const float uw = acc[eu / 16][eu % 16], rw = acc[er / 16][er % 16];
            yv[ol] = rw * sl + uw * S;
            us[2 * ol] = uw; us[2 * ol + 1] = S;
        }
        if constexpr (COAL && 2 * C::OH == 16) {
            // (uw, S) pairs: a pixel's 2 * O = 32 floats are one 128-byte row and the wave's 32 pixels 4 KB of consecutive
            // memory; through the wave's transposer they leave as lane-linear 16-byte pieces
            if (a.save_us) {
                const long r0 = tile * 128 + wv * 32;
                const int nv = (N - r0) < 32 ? (int)(N - r0 > 0 ? N - r0 : 0) : 32;
                ksr_store16_co<float, false, true, FULL>(xb, a.save_us, rg.first + r0, nv, lane, us);
            }
        } else if (a.save_us && live) {
            // (uw, S) of this lane's OH outputs: 2 * OH consecutive floats
            float* sp = a.save_us + (n * C::O + h * C::OH) * 2;
            if constexpr ((2 * C::OH) % 4 == 0) {
#pragma unroll
                for (int q = 0; q < 2 * C::OH / 4; ++q)
                    reinterpret_cast<float4*>(sp)[q] = make_float4(us[4 * q], us[4 * q + 1], us[4 * q + 2], us[4 * q + 3]);
            } else {
#pragma unroll
                for (int q = 0; q < 2 * C::OH; ++q) sp[q] = us[q];
            }
        }
        if (live) {
            // lane half h owns outputs o = h*OH + ol  ->  rows h*(R/2) + ol/R, columns ol%R
            float* op = a.out + (long)b * a.out_bstride + ((long)(y * R + h * (R / 2))) * (a.W * R) + (long)x * R;
#pragma unroll
            for (int rr = 0; rr < R / 2; ++rr) {
                if (R == 4)
                    *reinterpret_cast<float4*>(op + (long)rr * a.W * R) =
                        make_float4(yv[rr * R + 0], yv[rr * R + 1], yv[rr * R + 2], yv[rr * R + 3]);
                else
                    *reinterpret_cast<float2*>(op + (long)rr * a.W * R) = make_float2(yv[rr * R + 0], yv[rr * R + 1]);
            }
        }
    };
    if (COAL && (long)blockIdx.x < ntiles) issue_tile(blockIdx.x, 0);
    const long nfull = N / 128;                      // tiles [0, nfull) are complete; at most one ragged tile follows
    long tile = blockIdx.x;
    int it = 0;
    for (; tile < nfull; tile += gridDim.x, ++it) body(tile, it, std::true_type{});
    if (tile < ntiles) body(tile, it, std::false_type{});
    if constexpr (COAL) ksr_dma_wait_but<0>();       // no transfer outlives the wave (none is pending: the last tile requests nothing)
}

// Grid of the feed kernels: tiles over all pixels, or (tiles of one image, image) with per-image weights
dim3 feed_grid(const FeedArgs& a, long cap) {
    if (cap <= 0) cap = 512;
    if (a.w_bstride == 0) {
        const long ntiles = ((long)a.B * a.H * a.W + 127) / 128;
        return dim3((unsigned)(ntiles < cap ? ntiles : cap));
    }
    const long ntiles = ((long)a.H * a.W + 127) / 128;
    long per = (cap + a.B - 1) / a.B;
    if (per < 1) per = 1;
    return dim3((unsigned)(ntiles < per ? ntiles : per), (unsigned)a.B);
}

template <int R, int KC, int KORD>
int launch_feed_fwd(const FeedArgs& a, const KsrKnots& kn, hipStream_t s) {
    // one round of persistent workgroups (bf16 storage: 73 KB of LDS, two workgroups per CU: 768 ran as 512 + 256, half the
    // chip idle for the second half of the kernel)
#define KSR_CALL(P)                                                                                                      \
    do {                                                                                                                 \
        const dim3 grid = feed_grid(a, ksr_wg_slots(reinterpret_cast<const void*>(feed_kan_fwd_kernel<P, R, KC, KORD>), 256)); \
        hipLaunchKernelGGL((feed_kan_fwd_kernel<P, R, KC, KORD>), grid, dim3(256), 0, s, a, kn);                         \
    } while (0)
    // fp32 mode: the contraction as six bf16 matrix instructions on three-term splits where the three weight planes fit
    // the LDS (P <= 336: 132 KB), unless the caller turned that off
    if (ksr_prec() == KSR_F32 && ksr_f32_split() && ksr_frag_floats_p<KSR_F32X3>(32, FeedCfg<R, KC>::NB) * 4 <= 140 * 1024) KSR_CALL(KSR_F32X3);
    else KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    KSR_LAUNCH_CHECK("feed_kan_fwd");
    return KSR_OK;
}

// Gradient of the KAN evaluation with respect to the pixel's own luma value (the reference's autograd gives it when the
// input image requires a gradient; conv_sep_kan_cells.py:136-156 through kan_layer.py:68-84,106-118 and bspline.py:4-28):
//   dx[n] = sum_o go[n,o] * (rw[o] silu'(x) + uw[o] sum_g B'_g(x) coef[o,g]),
// with the order-0 indicator held constant in B'.  The per-pixel parameters are recomputed (feed product) as in the forward.
template <int PREC, int R, int KC, int KORD>
__global__ void __launch_bounds__(256, 2) feed_kan_dx_kernel(FeedArgs a, KsrKnots kn, float* __restrict__ gx, long gx_bstride) {
    using C = FeedCfg<R, KC>;
    constexpr int NB = C::NB;
    using ST = typename KsrStore<PREC>::T;
    __shared__ FeedSmem<NB, false> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, 32, NB, false>(sm.frag, sm.biasfrag, a.w + (long)blockIdx.y * a.w_bstride, 64,
                                           a.bias + (long)blockIdx.y * a.bias_bstride, MapKan{C::O, KC}, MapHL{32}, tid, 256);
    if (tid < KSR_MAX_KNOTS) sm.knots[tid] = kn.t[tid];
    __syncthreads();
    const int HW = a.H * a.W;
    const FeedRange rg = feed_range(a);              // all pixels, or those of image blockIdx.y (per-image weights)
    const long ntiles = (rg.count + 127) / 128;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r = tile * 128 + wv * 32 + tok;
        const bool live = r < rg.count;
        const long n = rg.first + r;
        const long nn = live ? n : rg.first;
        const int b = (int)(nn / HW), hw = (int)(nn % HW);
        const int y = hw / a.W, x = hw % a.W;
        float f[32];
        if (live) ksr_loadn<ST, 32>(a.feat, n * a.feat_ld + a.feat_c0 + 32 * h, f);
        else {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = 0.f;
        }
        f32x16 acc[NB];
        ksr_wave_gemm<PREC, 32, NB>(sm.frag, sm.biasfrag, f, acc, lane);
        const float xv = live ? a.x_img[(long)b * a.img_bstride + hw] : 0.f;
        float Nl[KORD + 1], dNl[KORD + 1], dNv[KSR_MAX_ORDER + 1], dbv[KC];
        const int m = ksr_bspline_local<KORD, true>(xv, sm.knots, kn.M, Nl, dNl);
#pragma unroll
        for (int q = 0; q <= KSR_MAX_ORDER; ++q) dNv[q] = 0.f;
#pragma unroll
        for (int q = 0; q <= KORD; ++q) dNv[q] = dNl[q];
        dense_basis<KC>(m, dNv, KORD, dbv);
        const float sg = ksr_sigmoid(xv), dsl = sg * (1.f + xv * (1.f - sg));
        float go[C::OH];
        {
            const float* gp = a.gout + (long)b * a.out_bstride + ((long)(y * R + h * (R / 2))) * (a.W * R) + (long)x * R;
#pragma unroll
            for (int rr = 0; rr < R / 2; ++rr)
#pragma unroll
                for (int j = 0; j < R; ++j) go[rr * R + j] = live ? gp[(long)rr * a.W * R + j] : 0.f;
        }
        float d = 0.f;
#pragma unroll
        for (int ol = 0; ol < C::OH; ++ol) {
            float dS = 0.f;
#pragma unroll
            for (int g = 0; g < KC; ++g) {
                const int e = ol * C::PER + g;
                dS += dbv[g] * acc[e / 16][e % 16];
            }
            const int eu = ol * C::PER + KC, er = eu + 1;
            const float uw = acc[eu / 16][eu % 16], rw = acc[er / 16][er % 16];
            d += go[ol] * (rw * dsl + uw * dS);
        }
        d = ksr_pair_sum(d);
        if (live && h == 0) gx[(long)b * gx_bstride + hw] = d;
    }
}

template <int R, int KC, int KORD>
int launch_feed_dx(const FeedArgs& a, const KsrKnots& kn, float* gx, long gx_bstride, hipStream_t s) {
    const dim3 grid = feed_grid(a, 768);
#define KSR_CALL(P) hipLaunchKernelGGL((feed_kan_dx_kernel<P, R, KC, KORD>), grid, dim3(256), 0, s, a, kn, gx, gx_bstride)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    KSR_LAUNCH_CHECK("feed_kan_dx");
    return KSR_OK;
}

template <int NB, bool COAL>
struct __attribute__((aligned(16))) FeedBwdSmem {
    float frag[2 * (NB * 16) * 64];
    float knots[KSR_MAX_KNOTS];
    float lnf[64];                                     // gain of the fused final LayerNorm
    // bf16 storage: per-wave tile through which the two outputs leave in lane-linear 16-byte pieces (a lane's own 160
    // and 64 contiguous bytes, stored directly, make every store instruction touch 64 separate segments); between the
    // two it serves as the wave's fp32 transposer for the fused final LayerNorm (hence at least KSR_XPOSE_FLOATS floats)
    static constexpr int OUT_TILE = 32 * (NB * 16 + 8 > 72 ? NB * 16 + 8 : 72);
    static constexpr int WAVE_TILE = OUT_TILE > 2 * KSR_XPOSE_FLOATS ? OUT_TILE : 2 * KSR_XPOSE_FLOATS;
    __bf16 tile[COAL ? 4 * WAVE_TILE : 8];
};

// Backward: dparam[e] per lane (lane order), dfeat = dparam * W_feed, both stored; the feed weight
// gradient is a separate reduction over pixels (ksr_wgrad with the KAN row map).
template <int PREC, int R, int KC, int KORD>
__global__ void __launch_bounds__(256, 2) feed_kan_bwd_kernel(FeedArgs a, KsrKnots kn) {
    using C = FeedCfg<R, KC>;
    constexpr int NB = C::NB, KH = NB * 16;
    using ST = typename KsrStore<PREC>::T;
    constexpr bool COAL = sizeof(ST) == 2;
    constexpr int TLD = KH + 8;                       // tile pitch for the parameter gradients (16 bytes of padding)
    constexpr int TILE = FeedBwdSmem<NB, COAL>::WAVE_TILE;   // elements per wave (the feature gradients use a pitch of 72)
    static_assert(!COAL || 32 * (TLD > 72 ? TLD : 72) <= TILE, "the output tiles fit");
    __shared__ FeedBwdSmem<NB, COAL> sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<PREC, KH, 2, true>(sm.frag, nullptr, a.w + (long)blockIdx.y * a.w_bstride, 64, nullptr, MapHL{32},
                                         MapKan{C::O, KC}, tid, 256);
    if (tid < KSR_MAX_KNOTS) sm.knots[tid] = kn.t[tid];
    const bool lnf = COAL && a.lnf_xhat != nullptr;
    if (lnf && tid < 64) sm.lnf[tid] = a.lnf_w[tid];
    float lcs0 = 0.f, lcs1 = 0.f;                    // lane = channel: sums of g * xhat and of g over this wave's tokens
    __syncthreads();

    const int HW = a.H * a.W;
    const FeedRange rg = feed_range(a);
    const long N = rg.first + rg.count;              // one past the last pixel of this workgroup's range
    const long ntiles = (rg.count + 127) / 128;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0 = rg.first + tile * 128 + wv * 32 + tok;
        const bool live = n0 < N;
        const long n = live ? n0 : rg.first;
        const int b = (int)(n / HW), hw = (int)(n % HW);
        const int y = hw / a.W, x = hw % a.W;

        float go[C::OH];
        {
            const float* gp = a.gout + (long)b * a.out_bstride + ((long)(y * R + h * (R / 2))) * (a.W * R) + (long)x * R;
#pragma unroll
            for (int rr = 0; rr < R / 2; ++rr) {
                if (R == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(gp + (long)rr * a.W * R);
                    go[rr * R + 0] = v.x; go[rr * R + 1] = v.y; go[rr * R + 2] = v.z; go[rr * R + 3] = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(gp + (long)rr * a.W * R);
                    go[rr * R + 0] = v.x; go[rr * R + 1] = v.y;
                }
            }
        }
        float us[C::OH * 2];
        const long n0w = rg.first + tile * 128 + wv * 32;            // the wave's 32 consecutive pixels
        __bf16* tl = sm.tile + (COAL ? wv * TILE : 0);
        if constexpr (COAL && 2 * C::OH == 16) {
            // the wave's 4 KB of (uw, S) pairs in lane-linear 16-byte pieces, transposed through the (still unused) tile
            static_assert(TILE * 2 >= 32 * KSR_XPOSE16_LD * 4, "the tile holds 32 rows of 36 floats");
            const int nv = (N - n0w) < 32 ? (int)(N - n0w > 0 ? N - n0w : 0) : 32;
            ksr_load16_co<float>(reinterpret_cast<float*>(tl), a.save_us, n0w, nv, lane, us);
        } else {
            const float4* up = reinterpret_cast<const float4*>(a.save_us + (n * C::O + h * C::OH) * 2);
#pragma unroll
            for (int q = 0; q < C::OH / 2; ++q) {
                const float4 v = up[q];
                us[4 * q + 0] = v.x; us[4 * q + 1] = v.y; us[4 * q + 2] = v.z; us[4 * q + 3] = v.w;
            }
        }
        const float xv = a.x_img[(long)b * a.img_bstride + hw];
        float Nv[KSR_MAX_ORDER + 1], bv[KC];
        const int m = basis_any<KORD>(xv, sm.knots, kn.M, Nv);
        dense_basis<KC>(m, Nv, KORD, bv);
        const float sl = ksr_silu(xv);

        float dp[KH];
#pragma unroll
        for (int e = 0; e < KH; ++e) {
            if (e < C::NH) {
                const int ol = e / C::PER, gi = e % C::PER;
                const float uw = us[2 * ol], S = us[2 * ol + 1];
                dp[e] = gi < KC ? go[ol] * uw * bv[gi < KC ? gi : 0] : (gi == KC ? go[ol] * S : go[ol] * sl);
            } else {
                dp[e] = 0.f;
            }
        }
        if constexpr (COAL) {
            // rows of gparam are [half 0: KH | half 1: KH] per pixel: one lane half at a time through the tile
            static_assert(KH % 8 == 0, "16-byte pieces");
            constexpr int PCS = KH / 8, TOTAL = 32 * PCS;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                ksr_wave_fence();
                if (h == hh) {
#pragma unroll
                    for (int q = 0; q < PCS; ++q) *reinterpret_cast<uint4*>(tl + tok * TLD + 8 * q) = ksr_pack8(dp + 8 * q);
                }
                ksr_wave_fence();
#pragma unroll
                for (int i = 0; i < (TOTAL + 63) / 64; ++i) {
                    const int idx = i * 64 + lane, row = idx / PCS, pc = idx - row * PCS;
                    if (idx < TOTAL && n0w + row < N)
                        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.gparam) + ((n0w + row) * 2 + hh) * KH + 8 * pc) =
                            *reinterpret_cast<const uint4*>(tl + row * TLD + 8 * pc);
                }
            }
        } else {
            if (live) ksr_storen<ST, KH>(a.gparam, (n * 2 + h) * KH, dp);
        }
        f32x16 acc[2];
        ksr_wave_gemm<PREC, KH, 2>(sm.frag, nullptr, dp, acc, lane);
        float df[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) { df[i] = acc[0][i]; df[16 + i] = acc[1][i]; }
        if constexpr (COAL) {
            if (lnf) {
                // backward of the LayerNorm that produced feat: column sums for its gain / shift, then the gradient of its input
                float* xbf = reinterpret_cast<float*>(tl);       // the parameter-gradient stores have read the tile (fence inside)
                const int nv = (N - n0w) < 32 ? (int)(N - n0w > 0 ? N - n0w : 0) : 32;
                float xh[32];
                ksr_load32_co<ST>(xbf, a.lnf_xhat, n0w, nv, 64, 0, lane, xh);
                if (!live) {
#pragma unroll
                    for (int i = 0; i < 32; ++i) df[i] = 0.f;
                }
                ksr_colsum32(xbf, lane, df, xh, lcs0, lcs1);
                const float rs = a.lnf_rstd[n];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) { df[i] *= sm.lnf[32 * h + i]; s1 += df[i]; s2 += df[i] * xh[i]; }
                const float m1 = ksr_pair_sum(s1) * (1.f / 64.f), m2 = ksr_pair_sum(s2) * (1.f / 64.f);
#pragma unroll
                for (int i = 0; i < 32; ++i) df[i] = rs * (df[i] - m1 - xh[i] * m2);
            }
            constexpr int FLD = 72;                       // 64 channels + 16 bytes of padding
            static_assert(32 * FLD <= TILE, "the feature-gradient tile fits");
            ksr_wave_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(tl + tok * FLD + 32 * h + 8 * q) = ksr_pack8(df + 8 * q);
            ksr_wave_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane, row = idx >> 3, pc = idx & 7;
                if (n0w + row < N)
                    *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.gfeat) + (n0w + row) * a.gfeat_ld + a.gfeat_c0 + 8 * pc) =
                        *reinterpret_cast<const uint4*>(tl + row * FLD + 8 * pc);
            }
        } else {
            if (live) ksr_storen<ST, 32>(a.gfeat, n * a.gfeat_ld + a.gfeat_c0 + 32 * h, df);
        }
    }
    if constexpr (COAL) {
        if (lnf) {
            // the four waves' sums meet in LDS (each wave's own tile first), one row [2][64] per workgroup
            const float* t0 = reinterpret_cast<const float*>(sm.tile);
            constexpr int WT = TILE / 2;                  // floats per wave tile
            float* xbf = reinterpret_cast<float*>(sm.tile + wv * TILE);
            ksr_wave_fence();
            xbf[lane] = lcs0; xbf[64 + lane] = lcs1;
            __syncthreads();
            for (int i = tid; i < 128; i += 256)
                a.lnf_slab[((long)blockIdx.y * gridDim.x + blockIdx.x) * 128 + i] = (t0[i] + t0[WT + i]) + (t0[2 * WT + i] + t0[3 * WT + i]);
        }
    }
}

template <int R, int KC, int KORD>
int launch_feed_bwd(const FeedArgs& a, const KsrKnots& kn, hipStream_t s, int* grid_out = nullptr) {
#define KSR_CALL(P)                                                                                                      \
    do {                                                                                                                 \
        long cap = ksr_wg_slots(reinterpret_cast<const void*>(feed_kan_bwd_kernel<P, R, KC, KORD>), 256);               \
        if (a.lnf_xhat && cap > 768) cap = 768;                    /* rows of the LayerNorm slab */                      \
        const dim3 grid = feed_grid(a, cap);                                                                             \
        if (grid_out) *grid_out = (int)(grid.x * grid.y);                                                                \
        hipLaunchKernelGGL((feed_kan_bwd_kernel<P, R, KC, KORD>), grid, dim3(256), 0, s, a, kn);                         \
    } while (0)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    KSR_LAUNCH_CHECK("feed_kan_bwd");
    return KSR_OK;
}

}  // namespace

// Fused kernels exist for upscale 2 or 4 with cubic splines (the reference's default spline_order) and grid_size in
// {3, 5, 8, 10, 16}, i.e. n_coef = grid_size + spline_order in {6, 8, 11, 13, 19}, and with grid_size 5 for the
// linear and quadratic splines (n_coef 6 and 7).
#define KSR_FEED_CONFIGS(X)                                                                                   \
    X(2, 6, 3) X(2, 8, 3) X(2, 11, 3) X(2, 13, 3) X(2, 19, 3) X(4, 6, 3) X(4, 8, 3) X(4, 11, 3) X(4, 13, 3) X(4, 19, 3) \
    X(2, 6, 1) X(2, 7, 2) X(4, 6, 1) X(4, 7, 2)

extern "C" int ksr_feed_kan_supported(int upscale, int n_coef, int spline_order) {
#define KSR_X(R, KC, KO) if (upscale == R && n_coef == KC && spline_order == KO) return 1;
    KSR_FEED_CONFIGS(KSR_X)
#undef KSR_X
    return 0;
}

static int ksr_feed_kan_P(int upscale, int n_coef) { return upscale * upscale * (n_coef + 2); }

static int feed_kan_fwd_impl(const float* feat, int feat_ld, int feat_c0, const float* x_img, int64_t img_batch_stride,
                             const float* feed_w, int per_image_weights, const float* feed_b, float* out,
                             int64_t out_batch_stride, float* save_us, int B, int H, int W, int upscale, int n_coef,
                             const float* knots, int n_knots, int spline_order, hipStream_t stream) {
    KSR_CHECK_ARG(feat && x_img && feed_w && feed_b && out && knots, "feed_kan_fwd: null pointer");
    KSR_CHECK_ARG(feat_ld >= 64 && feat_c0 >= 0 && feat_c0 + 64 <= feat_ld && feat_ld % 8 == 0 && feat_c0 % 8 == 0,
                  "feed_kan_fwd: bad feature row layout (ld %d, first channel %d)", feat_ld, feat_c0);
    KSR_CHECK_ARG(!per_image_weights || B <= 65535, "feed_kan_fwd: per-image weights take at most 65535 images");
    KSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "feed_kan_fwd: non-positive dimension");
    KSR_CHECK_ARG(n_knots == n_coef + spline_order + 1 && n_knots <= KSR_MAX_KNOTS, "feed_kan_fwd: bad knot count %d", n_knots);
    KSR_UNSUPPORTED(!ksr_feed_kan_supported(upscale, n_coef, spline_order),
                    "feed_kan_fwd: (upscale=%d, grid_size+spline_order=%d, spline_order=%d) has no fused kernel",
                    upscale, n_coef, spline_order);
    KSR_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0 && out_batch_stride % 4 == 0, "feed_kan_fwd: out must be 16-byte aligned");
    FeedArgs a{};
    a.feat = feat; a.x_img = x_img; a.img_bstride = img_batch_stride; a.w = feed_w; a.bias = feed_b;
    a.out = out; a.out_bstride = out_batch_stride; a.save_us = save_us; a.B = B; a.H = H; a.W = W;
    a.feat_ld = feat_ld; a.feat_c0 = feat_c0; a.w_bstride = per_image_weights ? (long)ksr_feed_kan_P(upscale, n_coef) * 64 : 0;
    a.bias_bstride = 0;
    KsrKnots kn;
    for (int i = 0; i < KSR_MAX_KNOTS; ++i) kn.t[i] = knots[i < n_knots ? i : n_knots - 1];
    kn.M = n_knots; kn.k = spline_order;
#define KSR_X(R, KC, KO) if (upscale == R && n_coef == KC && spline_order == KO) return launch_feed_fwd<R, KC, KO>(a, kn, stream);
    KSR_FEED_CONFIGS(KSR_X)
#undef KSR_X
    return KSR_ERR_UNSUPPORTED;
}

extern "C" int ksr_feed_kan_fwd(const float* feat, const float* x_img, int64_t img_batch_stride, const float* feed_w,
                                const float* feed_b, float* out, int64_t out_batch_stride, float* save_us, int B,
                                int H, int W, int upscale, int n_coef, const float* knots, int n_knots,
                                int spline_order, hipStream_t stream) {
    return feed_kan_fwd_impl(feat, 64, 0, x_img, img_batch_stride, feed_w, 0, feed_b, out, out_batch_stride, save_us, B, H, W,
                             upscale, n_coef, knots, n_knots, spline_order, stream);
}

// The same with one weight matrix PER IMAGE (feed_w [B][P][64]; the bias is shared) and the 64 features of a pixel at
// channels feat_c0 .. feat_c0 + 63 of rows of feat_ld elements: the channel-attention head, whose last 1x1 map is the
// per-image matrix M_b = W_a softmax(q^ k^T) applied to v (conv_sep_kan_attention.py:62-69), evaluated without the
// [N, P] parameter tensor.
extern "C" int ksr_feed_kan_fwd_img(const float* feat, int feat_ld, int feat_c0, const float* x_img, int64_t img_batch_stride,
                                    const float* feed_w_per_image, const float* feed_b, float* out, int64_t out_batch_stride,
                                    float* save_us, int B, int H, int W, int upscale, int n_coef, const float* knots,
                                    int n_knots, int spline_order, hipStream_t stream) {
    return feed_kan_fwd_impl(feat, feat_ld, feat_c0, x_img, img_batch_stride, feed_w_per_image, 1, feed_b, out, out_batch_stride,
                             save_us, B, H, W, upscale, n_coef, knots, n_knots, spline_order, stream);
}

// Per-half padded length of the lane-ordered parameter gradient: gparam is [N, 2, this].
extern "C" int ksr_feed_kan_lane_len(int upscale, int n_coef) {
    const int nh = (upscale * upscale / 2) * (n_coef + 2);
    return (nh + 15) / 16 * 16;
}

// gradient sums of up to three LayerNorms from per-workgroup rows [G][2][64] (ksr_tx_bwd.hip)
int ksr_ln_reduce(const float* slab, int G, float* out0, float* out1, float* out2, int row_stride, hipStream_t stream);

static int feed_kan_bwd_impl(const float* gout, int64_t gout_batch_stride, const float* x_img, int64_t img_batch_stride,
                             const float* feed_w, int per_image_weights, const float* save_us, float* gfeat, int gfeat_ld,
                             int gfeat_c0, float* gparam, int B, int H, int W, int upscale, int n_coef, const float* knots,
                             int n_knots, int spline_order, hipStream_t stream, const float* lnf_xhat = nullptr,
                             const float* lnf_rstd = nullptr, const float* lnf_w = nullptr, float* lnf_slab = nullptr,
                             float* lnf_grad = nullptr, int lnf_row_stride = 0) {
    KSR_CHECK_ARG(gout && x_img && feed_w && save_us && gfeat && gparam && knots, "feed_kan_bwd: null pointer");
    KSR_CHECK_ARG(gfeat_ld >= 64 && gfeat_c0 >= 0 && gfeat_c0 + 64 <= gfeat_ld && gfeat_ld % 8 == 0 && gfeat_c0 % 8 == 0,
                  "feed_kan_bwd: bad feature-gradient row layout (ld %d, first channel %d)", gfeat_ld, gfeat_c0);
    KSR_CHECK_ARG(!per_image_weights || B <= 65535, "feed_kan_bwd: per-image weights take at most 65535 images");
    KSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "feed_kan_bwd: non-positive dimension");
    KSR_CHECK_ARG(n_knots == n_coef + spline_order + 1 && n_knots <= KSR_MAX_KNOTS, "feed_kan_bwd: bad knot count %d", n_knots);
    KSR_UNSUPPORTED(!ksr_feed_kan_supported(upscale, n_coef, spline_order), "feed_kan_bwd: configuration has no fused kernel");
    KSR_CHECK_ARG((reinterpret_cast<uintptr_t>(gout) & 15) == 0 && gout_batch_stride % 4 == 0, "feed_kan_bwd: gout must be 16-byte aligned");
    FeedArgs a{};
    a.gout = gout; a.out_bstride = gout_batch_stride; a.x_img = x_img; a.img_bstride = img_batch_stride;
    a.w = feed_w; a.save_us = const_cast<float*>(save_us); a.gfeat = gfeat; a.gparam = gparam; a.B = B; a.H = H; a.W = W;
    a.gfeat_ld = gfeat_ld; a.gfeat_c0 = gfeat_c0; a.feat_ld = 64;
    a.w_bstride = per_image_weights ? (long)ksr_feed_kan_P(upscale, n_coef) * 64 : 0;
    if (lnf_xhat) {
        KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "feed_kan_bwd: the fused LayerNorm backward exists in the bf16 storage mode only");
        KSR_CHECK_ARG(lnf_rstd && lnf_w && lnf_slab && lnf_grad && lnf_row_stride >= 64 && gfeat_ld == 64 && gfeat_c0 == 0,
                      "feed_kan_bwd: the fused LayerNorm backward needs rstd, gain, a slab, its gradient pair and a dense gfeat");
        a.lnf_xhat = lnf_xhat; a.lnf_rstd = lnf_rstd; a.lnf_w = lnf_w; a.lnf_slab = lnf_slab;
    }
    KsrKnots kn;
    for (int i = 0; i < KSR_MAX_KNOTS; ++i) kn.t[i] = knots[i < n_knots ? i : n_knots - 1];
    kn.M = n_knots; kn.k = spline_order;
    int grid = 0, rc = KSR_ERR_UNSUPPORTED;
#define KSR_X(R, KC, KO) if (upscale == R && n_coef == KC && spline_order == KO) rc = launch_feed_bwd<R, KC, KO>(a, kn, stream, &grid);
    KSR_FEED_CONFIGS(KSR_X)
#undef KSR_X
    if (rc != KSR_OK || !lnf_xhat) return rc;
    return ksr_ln_reduce(lnf_slab, grid, lnf_grad, nullptr, nullptr, lnf_row_stride, stream);
}

extern "C" int ksr_feed_kan_bwd(const float* gout, int64_t gout_batch_stride, const float* x_img,
                                int64_t img_batch_stride, const float* feed_w, const float* save_us, float* gfeat,
                                float* gparam, int B, int H, int W, int upscale, int n_coef, const float* knots,
                                int n_knots, int spline_order, hipStream_t stream) {
    return feed_kan_bwd_impl(gout, gout_batch_stride, x_img, img_batch_stride, feed_w, 0, save_us, gfeat, 64, 0, gparam, B, H, W,
                             upscale, n_coef, knots, n_knots, spline_order, stream);
}

// ksr_feed_kan_bwd with the backward of the LayerNorm that produced feat (the encoder's final norm, gazetr.py:113) inside
// (bf16 storage mode): gfeat receives the gradient of that LayerNorm's INPUT, ln_grad its gain gradient at [c] and its shift
// gradient at [ln_row_stride + c].  xhat [N,64], rstd [N]: saved by the forward; ln_slab: 768 * 128 floats.
extern "C" int ksr_feed_kan_bwd_lnf(const float* gout, int64_t gout_batch_stride, const float* x_img, int64_t img_batch_stride,
                                    const float* feed_w, const float* save_us, float* gfeat, float* gparam, int B, int H, int W,
                                    int upscale, int n_coef, const float* knots, int n_knots, int spline_order, const float* xhat,
                                    const float* rstd, const float* ln_w, float* ln_slab, float* ln_grad, int ln_row_stride,
                                    hipStream_t stream) {
    KSR_CHECK_ARG(xhat != nullptr, "feed_kan_bwd_lnf: null xhat");
    return feed_kan_bwd_impl(gout, gout_batch_stride, x_img, img_batch_stride, feed_w, 0, save_us, gfeat, 64, 0, gparam, B, H, W,
                             upscale, n_coef, knots, n_knots, spline_order, stream, xhat, rstd, ln_w, ln_slab, ln_grad, ln_row_stride);
}

// Backward of ksr_feed_kan_fwd_img: per-image weights; the feature gradient goes to channels gfeat_c0 .. + 63 of rows of
// gfeat_ld elements (e.g. the v third of a [N,192] gradient tensor).
extern "C" int ksr_feed_kan_bwd_img(const float* gout, int64_t gout_batch_stride, const float* x_img, int64_t img_batch_stride,
                                    const float* feed_w_per_image, const float* save_us, float* gfeat, int gfeat_ld, int gfeat_c0,
                                    float* gparam, int B, int H, int W, int upscale, int n_coef, const float* knots,
                                    int n_knots, int spline_order, hipStream_t stream) {
    return feed_kan_bwd_impl(gout, gout_batch_stride, x_img, img_batch_stride, feed_w_per_image, 1, save_us, gfeat, gfeat_ld,
                             gfeat_c0, gparam, B, H, W, upscale, n_coef, knots, n_knots, spline_order, stream);
}

// gx[b, y, x] (images gx_batch_stride elements apart) = d(out)/d(luma) contracted with gout, through the KAN evaluation's
// direct dependence on the pixel's luma value (the path through the encoder is ksr_stem_dx).  Written, not accumulated.
namespace {
int feed_kan_dx_impl(const float* feat, int feat_ld, int feat_c0, const float* x_img, int64_t img_batch_stride, const float* feed_w,
                     int per_image, const float* feed_b, const float* gout, int64_t gout_batch_stride, float* gx,
                     int64_t gx_batch_stride, int B, int H, int W, int upscale, int n_coef, const float* knots, int n_knots,
                     int spline_order, hipStream_t stream) {
    KSR_CHECK_ARG(feat && x_img && feed_w && feed_b && gout && gx && knots, "feed_kan_dx: null pointer");
    KSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "feed_kan_dx: non-positive dimension");
    KSR_CHECK_ARG(!per_image || B <= 65535, "feed_kan_dx: at most 65535 images with per-image weights");
    const int al = ksr_prec() == KSR_BF16S ? 8 : 4;        // elements per 16-byte piece of a feature row
    KSR_CHECK_ARG(feat_ld >= 64 && feat_c0 >= 0 && feat_c0 + 64 <= feat_ld && feat_ld % al == 0 && feat_c0 % al == 0,
                  "feed_kan_dx: bad feature row layout (ld %d, c0 %d)", feat_ld, feat_c0);
    KSR_CHECK_ARG(n_knots == n_coef + spline_order + 1 && n_knots <= KSR_MAX_KNOTS, "feed_kan_dx: bad knot count %d", n_knots);
    KSR_UNSUPPORTED(!ksr_feed_kan_supported(upscale, n_coef, spline_order), "feed_kan_dx: configuration has no fused kernel");
    FeedArgs a{};
    a.feat = feat; a.x_img = x_img; a.img_bstride = img_batch_stride; a.w = feed_w; a.bias = feed_b;
    a.gout = gout; a.out_bstride = gout_batch_stride; a.B = B; a.H = H; a.W = W;
    a.feat_ld = feat_ld; a.feat_c0 = feat_c0;
    a.w_bstride = per_image ? (long)upscale * upscale * (n_coef + 2) * 64 : 0;     // the bias is shared
    KsrKnots kn;
    for (int i = 0; i < KSR_MAX_KNOTS; ++i) kn.t[i] = knots[i < n_knots ? i : n_knots - 1];
    kn.M = n_knots; kn.k = spline_order;
#define KSR_X(R, KC, KO) if (upscale == R && n_coef == KC && spline_order == KO) return launch_feed_dx<R, KC, KO>(a, kn, gx, (long)gx_batch_stride, stream);
    KSR_FEED_CONFIGS(KSR_X)
#undef KSR_X
    return KSR_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" int ksr_feed_kan_dx(const float* feat, const float* x_img, int64_t img_batch_stride, const float* feed_w,
                               const float* feed_b, const float* gout, int64_t gout_batch_stride, float* gx,
                               int64_t gx_batch_stride, int B, int H, int W, int upscale, int n_coef, const float* knots,
                               int n_knots, int spline_order, hipStream_t stream) {
    return feed_kan_dx_impl(feat, 64, 0, x_img, img_batch_stride, feed_w, 0, feed_b, gout, gout_batch_stride, gx, gx_batch_stride,
                            B, H, W, upscale, n_coef, knots, n_knots, spline_order, stream);
}

// The same with the per-image weights of ksr_feed_kan_fwd_img (and the features at channels feat_c0 .. of rows of feat_ld).
extern "C" int ksr_feed_kan_dx_img(const float* feat, int feat_ld, int feat_c0, const float* x_img, int64_t img_batch_stride,
                                   const float* feed_w_per_image, const float* feed_b, const float* gout,
                                   int64_t gout_batch_stride, float* gx, int64_t gx_batch_stride, int B, int H, int W,
                                   int upscale, int n_coef, const float* knots, int n_knots, int spline_order,
                                   hipStream_t stream) {
    return feed_kan_dx_impl(feat, feat_ld, feat_c0, x_img, img_batch_stride, feed_w_per_image, 1, feed_b, gout, gout_batch_stride,
                            gx, gx_batch_stride, B, H, W, upscale, n_coef, knots, n_knots, spline_order, stream);
}
