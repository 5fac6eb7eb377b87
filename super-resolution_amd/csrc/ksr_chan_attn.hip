// Channel attention (ConvSepKanAttention.forward, conv_sep_kan/conv_sep_kan_attention.py:48-71) on NHWC
// activations, 64 mid channels:
//     q, k, v = dw3x3(pw(x));  q^, k^ = q, k / max(||.||_HW, 1e-12);  A = softmax_c'(q^T k^);  y = a_proj(A v)
// The [N, 64] tensors are only ever touched by streaming kernels; everything that is 64 x 64 per image is
// folded into small per-image matrices:
//     forward   G_b = q^T k (batched Gram product, with the squared norms)  ->  A_b, S_b  ->  M_b = W_a A_b
//               y[n] = M_b v[n] + b_a                                   (per-image 1x1 conv, MFMA)
//     backward  Gd_b = dy^T v  ->  dA_b = W_a^T Gd_b, softmax and normalisation derivatives  ->
//               dv[n] = M_b^T dy[n];  [dq | dk][n] = Mqk_b [q | k][n];  dW_a = sum_b Gd_b A_b^T
// so neither A v nor its gradient is materialised.
#include "ksr_conv_common.h"

namespace {

// ---- depthwise 3x3 over `groups` blocks of 64 channels (FLIP: transposed conv, no bias) -------------
struct DwArgs {
    const float* in; const float* dw_w; const float* dw_b; float* out;
    int ld, groups, B, H, W;
};

template <class ST, bool FLIP>      // ST: storage type of in / out (float, or __bf16 in the bf16-storage mode)
__global__ void __launch_bounds__(256, 2) chan_dw_kernel(DwArgs a) {
    __shared__ __attribute__((aligned(16))) float halo[HALO_FLOATS];
    __shared__ __attribute__((aligned(16))) float dww[3 * 9 * 64];
    __shared__ __attribute__((aligned(16))) float dwb[3 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    for (int gi = 0; gi < a.groups; ++gi) {
        ksr_stage_dw<FLIP>(dww + gi * 576, a.dw_w + gi * 576, tid);
        if (tid < 64) dwb[gi * 64 + tid] = (FLIP || !a.dw_b) ? 0.f : a.dw_b[gi * 64 + tid];
    }
    __syncthreads();
    const int HW = a.H * a.W;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    // bf16 tensors: the halo of the next (tile, group) is requested before the current one's arithmetic (without it every
    // group of every tile exposed one full memory latency: 806 MB per launch moved at 3.4 TB/s)
    constexpr bool HAHEAD = sizeof(ST) == 2;
    KsrHalo<ST> hr;
    auto fetch = [&](long t, int gi) {
        const int ftx = (int)(t % tiles_x), fty = (int)((t / tiles_x) % tiles_y);
        const int fb = (int)(t / ((long)tiles_x * tiles_y));
        ksr_halo_fetch<ST>(a.in, a.ld, 64 * gi, fb, fty * 4, ftx * 32, a.H, a.W, tid, hr);
    };
    const KsrTileWalk tw = ksr_xcd_tile_walk(ntiles);
    if (HAHEAD && tw.first < tw.end) fetch(tw.first, 0);
    for (long tile = tw.first; tile < tw.end; tile += tw.step) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const int b = (int)(tile / ((long)tiles_x * tiles_y));
        const int y0 = ty * 4, x0 = tx * 32;
        const int y = y0 + wv;
        for (int gi = 0; gi < a.groups; ++gi) {
            __syncthreads();
            if (!HAHEAD) fetch(tile, gi);
            ksr_halo_commit<ST>(halo, tid, hr);
            __syncthreads();
            if (HAHEAD) {
                if (gi + 1 < a.groups) fetch(tile, gi + 1);
                else if (tile + tw.step < tw.end) fetch(tile + tw.step, 0);
            }
            float g[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) g[i] = dwb[gi * 64 + 32 * h + i];
            ksr_dwconv_lane(halo, dww + gi * 576, wv, tok, h, g);
            if constexpr (sizeof(ST) == 4) {
                // fp32 tensors: the wave's 32-pixel row leaves in lane-linear 16-byte pieces through the halo tile, which
                // serves as the per-wave transposers once every wave is done with it (a lane's own 128 contiguous bytes,
                // stored directly, make each of eight store instructions touch 64 separate segments; measured 17.8 ->
                // 17.4 ms on the fp32 step).  bf16 tensors (64 bytes per lane) are faster stored directly (0.23 against
                // 0.26 ms per launch).
                __syncthreads();
                const long n0w = (long)b * HW + (long)y * a.W + x0;
                const int nvalid = (y < a.H) ? (a.W - x0 < 32 ? a.W - x0 : 32) : 0;
                ksr_store32_co<ST>(halo + wv * KSR_XPOSE_FLOATS, a.out, n0w, nvalid, a.ld, 64 * gi, lane, g);
            } else {
                const int x = x0 + tok;
                if (y < a.H && x < a.W) ksr_storen<ST, 32>(a.out, ((long)b * HW + (long)y * a.W + x) * a.ld + 64 * gi + 32 * h, g);
            }
        }
    }
}

// ---- transposed depthwise 3x3 of q | k | v with the data gradient of their 1x1 projections in the epilogue (bf16 storage) ----
// din [N,192] = convT(dout) as chan_dw_kernel<__bf16, true>, and dx [N,64] = din W (W [192][64]: the three projections'
// weights, q | k | v): a lane's 32 results per group are already the token-per-lane operand of ksr_wave_gemm (half h holds
// channels 64 g + 32 h + i: MapHL{96}, the layout ksr_pointwise_bwd's kernel uses, so the sums keep its order), which
// saves that kernel's pass over din (402 MB per launch at 64 x 128 x 128 pixels).  LDS: the halo tile, the 192 x 64 weight
// fragments (24 KB) and nothing else - the depthwise table of the current group lives in the halo tile's padding and is
// restaged with every group's halo (its 576 weights are requested together with that halo, one (tile, group) ahead).
struct DwDxArgs { const float* dout; const float* dw_w; const float* pw_w; float* din; float* dx; int B, H, W; };

__global__ void __launch_bounds__(256, 2) chan_dw_bwd_dx_kernel(DwDxArgs a) {
    using ST = __bf16;
    __shared__ __attribute__((aligned(16))) float halo[HALO_FLOATS];
    __shared__ __attribute__((aligned(16))) float frag[ksr_frag_floats_p<KSR_BF16S>(96, 2)];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    ksr_stage_weights<KSR_BF16S, 96, 2, true>(frag, nullptr, a.pw_w, 64, nullptr, MapHL{32}, MapHL{96}, tid, 256);
    const int HW = a.H * a.W;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    KsrHalo<ST> hr;
    float twr[3];
    auto fetch = [&](long t, int gi) {
        const int ftx = (int)(t % tiles_x), fty = (int)((t / tiles_x) % tiles_y);
        const int fb = (int)(t / ((long)tiles_x * tiles_y));
        ksr_halo_fetch<ST>(a.dout, 192, 64 * gi, fb, fty * 4, ftx * 32, a.H, a.W, tid, hr);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + 256 * k;                // element i = tap * 64 + channel of the [tap][channel] table, taps flipped
            twr[k] = i < 9 * 64 ? a.dw_w[gi * 576 + (i & 63) * 9 + (8 - (i >> 6))] : 0.f;
        }
    };
    const KsrTileWalk tw = ksr_xcd_tile_walk(ntiles);
    if (tw.first < tw.end) fetch(tw.first, 0);
    for (long tile = tw.first; tile < tw.end; tile += tw.step) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const int b = (int)(tile / ((long)tiles_x * tiles_y));
        const int y0 = ty * 4, x0 = tx * 32;
        const int y = y0 + wv, x = x0 + tok;
        float xall[96];
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            __syncthreads();           // the previous group's readers (and the previous tile's transposers) are done
            ksr_halo_commit<ST>(halo, tid, hr);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = tid + 256 * k;
                if (i < 9 * 64) halo[(i >> 2) * HALO_LD + 64 + (i & 3)] = twr[k];
            }
            __syncthreads();
            if (gi + 1 < 3) fetch(tile, gi + 1);
            else if (tile + tw.step < tw.end) fetch(tile + tw.step, 0);
            float g[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) g[i] = 0.f;
            ksr_dwconv_lane_pad(halo, wv, tok, h, g);
            if (y < a.H && x < a.W) ksr_storen<ST, 32>(a.din, ((long)b * HW + (long)y * a.W + x) * 192 + 64 * gi + 32 * h, g);
#pragma unroll
            for (int i = 0; i < 32; ++i) xall[32 * gi + i] = g[i];
        }
        __syncthreads();               // every wave is done with the halo tile: its pixel rows serve as the transposers
        f32x16 acc[2];
        ksr_wave_gemm<KSR_BF16S, 96, 2>(frag, nullptr, xall, acc, lane);
        float o[32];
#pragma unroll
        for (int q = 0; q < 16; ++q) { o[q] = acc[0][q]; o[16 + q] = acc[1][q]; }
        const long n0w = (long)b * HW + (long)y * a.W + x0;
        const int nvalid = (y < a.H) ? (a.W - x0 < 32 ? a.W - x0 : 32) : 0;
        ksr_store32_co<ST>(halo + wv * KSR_XPOSE_FLOATS, a.dx, n0w, nvalid, 64, 0, lane, o);
    }
}

// ---- per-image 1x1 conv: out[n, :NOUT] = W_b(n) in[n, :KIN] + bias -----------------------------------
// Lane half h holds input channels [h*kin_h, (h+1)*kin_h) and produces outputs [h*nout_h, (h+1)*nout_h).
struct ApArgs {
    const float* in; int in_ld, in_c0, kin_h;
    const float* w; long w_bstride; int ldw;
    const float* bias;
    float* out; int out_ld, out_c0, nout_h;
    int HW;
};

template <int PREC, int KH, int NB, bool TRANS>
__global__ void __launch_bounds__(256) chan_apply_kernel(ApArgs a) {
    extern __shared__ __attribute__((aligned(16))) float apmem[];
    float* frag = apmem;
    float* biasfrag = apmem + NB * KH * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    using ST = typename KsrStore<PREC>::T;
    ksr_stage_weights<PREC, KH, NB, TRANS>(frag, biasfrag, a.w + (long)b * a.w_bstride, a.ldw, a.bias,
                                          MapSplit{a.nout_h}, MapSplit{a.kin_h}, tid, 256);
    __syncthreads();
    const int ntiles = (a.HW + 127) / 128;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int t = tile * 128 + wv * 32 + tok;
        const bool live = t < a.HW;
        const long n = (long)b * a.HW + (live ? t : 0);
        float x[KH];
        if constexpr (sizeof(ST) == 4) {
            const float4* src = reinterpret_cast<const float4*>(a.in + n * a.in_ld + a.in_c0 + h * a.kin_h);
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (4 * q < a.kin_h) v = src[q];
                x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
        } else {
            // bf16 storage: 16-byte pieces of 8 channels (the host checks that the halves are multiples of 8)
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.in) + n * a.in_ld + a.in_c0 + h * a.kin_h);
#pragma unroll
            for (int q = 0; q < KH / 8; ++q) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (8 * q < a.kin_h) v = src[q];
                ksr_unpack8(v, x + 8 * q);
            }
        }
        f32x16 acc[NB];
        ksr_wave_gemm<PREC, KH, NB>(frag, biasfrag, x, acc, lane);
        if (live) {
            if constexpr (sizeof(ST) == 4) {
                float4* dst = reinterpret_cast<float4*>(a.out + n * a.out_ld + a.out_c0 + h * a.nout_h);
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        if (16 * blk + 4 * q4 < a.nout_h)
                            dst[4 * blk + q4] = make_float4(acc[blk][4 * q4 + 0], acc[blk][4 * q4 + 1], acc[blk][4 * q4 + 2],
                                                            acc[blk][4 * q4 + 3]);
            } else {
                uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.out) + n * a.out_ld + a.out_c0 + h * a.nout_h);
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                    for (int q8 = 0; q8 < 2; ++q8)
                        if (16 * blk + 8 * q8 < a.nout_h) {
                            float o[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] = acc[blk][8 * q8 + j];
                            dst[2 * blk + q8] = ksr_pack8(o);
                        }
            }
        }
    }
}

template <int PREC, int KH, int NB, bool TRANS>
int launch_apply_p(const ApArgs& a, int B, hipStream_t s) {
    const size_t lds = (size_t)(NB * KH * 64 + NB * 32) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)chan_apply_kernel<PREC, KH, NB, TRANS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { ksr_set_error("chan_apply: hipFuncSetAttribute: %s", hipGetErrorString(e)); return KSR_ERR_HIP; }
        attr_set = true;
    }
    const int ntiles = (a.HW + 127) / 128;
    int gx = 1024 / B;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL((chan_apply_kernel<PREC, KH, NB, TRANS>), dim3(gx, B), dim3(256), lds, s, a);
    KSR_LAUNCH_CHECK("chan_apply");
    return KSR_OK;
}

template <int KH, int NB, bool TRANS>
int launch_apply(const ApArgs& a, int B, hipStream_t s) {
    if (ksr_prec() == KSR_BF16S) {
        KSR_UNSUPPORTED(a.kin_h % 8 || a.nout_h % 8 || a.in_ld % 8 || a.in_c0 % 8 || a.out_ld % 8 || a.out_c0 % 8,
                        "chan_apply: bf16 storage needs channel halves, row lengths and offsets that are multiples of 8");
        return launch_apply_p<KSR_BF16S, KH, NB, TRANS>(a, B, s);
    }
    return ksr_prec() == KSR_BF16 ? launch_apply_p<KSR_BF16, KH, NB, TRANS>(a, B, s) : launch_apply_p<KSR_F32, KH, NB, TRANS>(a, B, s);
}

// ---- per-image 64 x 64 algebra ----------------------------------------------------------------------
struct SoftArgs {
    const float* G;        // [B][64][64]  q^T k
    const float* tail;     // [B][192]     (sum q | sum q^2 | sum k^2)
    const float* Wa;       // [P][64]
    float* A;              // [B][64][64]  softmax
    float* S;              // [B][64][64]  normalised scores
    float* nrm;            // [B][128]     raw norms of q, k
    float* M;              // [B][P][64]   W_a A_b
    int P;
};

// grid (B, ny): every workgroup of an image forms the 64 x 64 softmax (cheap) and takes a chunk of M's rows; the first one
// also writes A, S and the norms.  (One workgroup per image left the P x 64 x 64 product to 4 waves on 64 of 256 CUs.)
__global__ void __launch_bounds__(256) chan_soft_fwd_kernel(SoftArgs a) {
    __shared__ float As[64][65];
    __shared__ float nq[64], nk[64];
    const int tid = threadIdx.x, b = blockIdx.x;
    const bool first = blockIdx.y == 0;
    if (tid < 128) {
        const float raw = sqrtf(a.tail[(long)b * 192 + 64 + tid]);
        if (first) a.nrm[(long)b * 128 + tid] = raw;
        const float cl = fmaxf(raw, 1e-12f);
        if (tid < 64) nq[tid] = cl; else nk[tid - 64] = cl;
    }
    __syncthreads();
    const int c = tid >> 2, part = tid & 3;
    const float* Gr = a.G + (long)b * 4096 + c * 64 + part * 16;
    float s[16];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        s[j] = Gr[j] / (nq[c] * nk[part * 16 + j]);
        mx = fmaxf(mx, s[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float e[16], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { e[j] = expf(s[j] - mx); sum += e[j]; }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float p = e[j] * inv;
        As[c][part * 16 + j] = p;
        if (first) {
            a.A[(long)b * 4096 + c * 64 + part * 16 + j] = p;
            a.S[(long)b * 4096 + c * 64 + part * 16 + j] = s[j];
        }
    }
    __syncthreads();
    // M = W_a A: a wave owns rows p = p_lo + wave + 4 k, lane = column cc; A's column stays in registers, W_a's row is read
    // through wave-uniform (scalar) loads; two rows at a time for two independent chains (each row's sum keeps its order)
    {
        const int chunk = (((a.P + (int)gridDim.y - 1) / (int)gridDim.y) + 7) & ~7;
        const int p_lo = (int)blockIdx.y * chunk, p_hi = p_lo + chunk < a.P ? p_lo + chunk : a.P;
        const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6), cc = tid & 63;
        float ar[64];
#pragma unroll
        for (int ci = 0; ci < 64; ++ci) ar[ci] = As[ci][cc];
        float* Mb = a.M + (long)b * a.P * 64;
        int p = p_lo + w0;
        for (; p + 4 < p_hi; p += 8) {
            const float* wa = a.Wa + p * 64;
            const float* wb = wa + 4 * 64;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int ci = 0; ci < 64; ++ci) { acc0 += wa[ci] * ar[ci]; acc1 += wb[ci] * ar[ci]; }
            Mb[p * 64 + cc] = acc0;
            Mb[(p + 4) * 64 + cc] = acc1;
        }
        for (; p < p_hi; p += 4) {
            const float* wa = a.Wa + p * 64;
            float acc0 = 0.f;
#pragma unroll
            for (int ci = 0; ci < 64; ++ci) acc0 += wa[ci] * ar[ci];
            Mb[p * 64 + cc] = acc0;
        }
    }
}

struct SoftBwdArgs {
    const float* Gd;       // [B][P][64]  dy^T v
    const float* A; const float* S; const float* nrm; const float* Wa;
    float* Mqk;            // [B][128][128]
    float* dWa_part;       // [B][P][64]
    int P;
};

// grid (B, 1 + K), 1024 threads: workgroup (b, 0) forms dS and Mqk of image b, workgroups (b, 1..K) a chunk each of the
// image's dW_a contribution (which needs only A and Gd).
__global__ void __launch_bounds__(1024) chan_soft_bwd_kernel(SoftBwdArgs a) {
    __shared__ float As[64][65], Ss[64][65], dS[64][65];
    __shared__ float nq[64], nk[64], pq[64], pk[64], r[64], rp[64];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* Gd = a.Gd + (long)b * a.P * 64;
    for (int idx = tid; idx < 4096; idx += 1024) As[idx >> 6][idx & 63] = a.A[(long)b * 4096 + idx];
    if (blockIdx.y > 0) {
        // dW_a contribution of this image: Gd A^T.  A wave owns rows p = p_lo + wave + 16 k, lane = column c: row c of A stays
        // in registers, Gd's row is read through wave-uniform (scalar) loads
        __syncthreads();
        const int K = (int)gridDim.y - 1;
        const int chunk = (((a.P + K - 1) / K) + 15) & ~15;
        const int p_lo = ((int)blockIdx.y - 1) * chunk, p_hi = p_lo + chunk < a.P ? p_lo + chunk : a.P;
        const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6), c = tid & 63;
        float ar[64];
#pragma unroll
        for (int cc = 0; cc < 64; ++cc) ar[cc] = As[c][cc];
        for (int p = p_lo + w0; p < p_hi; p += 16) {
            const float* g = Gd + p * 64;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int cc = 0; cc < 64; cc += 4) {
                s0 += g[cc] * ar[cc]; s1 += g[cc + 1] * ar[cc + 1]; s2 += g[cc + 2] * ar[cc + 2]; s3 += g[cc + 3] * ar[cc + 3];
            }
            a.dWa_part[(long)b * a.P * 64 + p * 64 + c] = (s0 + s1) + (s2 + s3);
        }
        return;
    }
    for (int idx = tid; idx < 4096; idx += 1024) Ss[idx >> 6][idx & 63] = a.S[(long)b * 4096 + idx];
    if (tid < 128) {
        const float raw = a.nrm[(long)b * 128 + tid];
        const float cl = fmaxf(raw, 1e-12f), pr = raw >= 1e-12f ? 1.f : 0.f;
        if (tid < 64) { nq[tid] = cl; pq[tid] = pr; } else { nk[tid - 64] = cl; pk[tid - 64] = pr; }
    }
    // dA = W_a^T Gd: a wave owns rows c = wave + 16 k (k < 4), lane = column; per p one coalesced read of Gd's row and four
    // wave-uniform (scalar) reads of W_a feed four independent sums; eight p in flight at a time (the loop is a chain of
    // memory latencies otherwise)
    {
        const int c0 = __builtin_amdgcn_readfirstlane(tid >> 6), cc = tid & 63;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int p = 0; p < a.P; ++p) {
            const float gv = Gd[p * 64 + cc];
            const float* wr = a.Wa + p * 64 + c0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += wr[16 * k] * gv;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dS[c0 + 16 * k][cc] = acc[k];
    }
    __syncthreads();
    if (tid < 256) {   // softmax derivative, row sums of dS * S
        const int c = tid >> 2, part = tid & 3;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) dot += dS[c][part * 16 + j] * As[c][part * 16 + j];
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        float rr = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cc = part * 16 + j;
            const float d = As[c][cc] * (dS[c][cc] - dot);
            dS[c][cc] = d;
            rr += d * Ss[c][cc];
        }
        rr += __shfl_xor(rr, 1, 64);
        rr += __shfl_xor(rr, 2, 64);
        if (part == 0) r[c] = rr;
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int c = 0; c < 64; ++c) s += dS[c][tid] * Ss[c][tid];
        rp[tid] = s;
    }
    __syncthreads();
    // [dq | dk] = Mqk [q | k]
    float* Mqk = a.Mqk + (long)b * 16384;
    for (int idx = tid; idx < 16384; idx += 1024) {
        const int o = idx >> 7, i = idx & 127;
        float v;
        if (o < 64) {
            if (i < 64) v = (i == o) ? -pq[o] * r[o] / (nq[o] * nq[o]) : 0.f;
            else v = dS[o][i - 64] / (nq[o] * nk[i - 64]);
        } else {
            const int cc = o - 64;
            if (i < 64) v = dS[i][cc] / (nq[i] * nk[cc]);
            else v = (i - 64 == cc) ? -pk[cc] * rp[cc] / (nk[cc] * nk[cc]) : 0.f;
        }
        Mqk[idx] = v;
    }
}

// out[i] = sum_g in[g][i]  (fixed order); four entries in flight per thread
__global__ void __launch_bounds__(256) sum_entries_kernel(const float* in, int G, long stride, int n, float* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
        const float v0 = in[(long)g * stride + i], v1 = in[(long)(g + 1) * stride + i];
        const float v2 = in[(long)(g + 2) * stride + i], v3 = in[(long)(g + 3) * stride + i];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; g < G; ++g) s += in[(long)g * stride + i];
    out[i] = s;
}

int dims_ok(int B, int H, int W, const char* who) {
    KSR_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "%s: bad dimensions", who);
    return KSR_OK;
}

}  // namespace

// q | k | v = dw3x3(pq) for `groups` (1..3) blocks of 64 channels of an NHWC tensor with `ld` channels per
// pixel (q_dw, k_dw, v_dw: conv_sep_kan_attention.py:16-37); dw_w [groups*64][9], dw_b [groups*64].
extern "C" int ksr_chan_dw_fwd(const float* in, const float* dw_w, const float* dw_b, float* out, int ld, int groups,
                               int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(in && dw_w && dw_b && out, "chan_dw_fwd: null pointer");
    const bool st16 = ksr_prec() == KSR_BF16S;      // in / out are bf16 tensors in the bf16-storage mode
    KSR_CHECK_ARG(groups >= 1 && groups <= 3 && ld >= 64 * groups && ld % (st16 ? 8 : 4) == 0, "chan_dw_fwd: bad channel layout");
    if (int rc = dims_ok(B, H, W, "chan_dw_fwd")) return rc;
    DwArgs a{in, dw_w, dw_b, out, ld, groups, B, H, W};
    const long ntiles = (long)B * ((H + 3) / 4) * ((W + 31) / 32);
    const dim3 grid((unsigned)(ntiles < 1024 ? ntiles : 1024));
    if (st16) hipLaunchKernelGGL((chan_dw_kernel<__bf16, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((chan_dw_kernel<float, false>), grid, dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("chan_dw_fwd");
    return KSR_OK;
}

// Transposed depthwise conv of the same layout: din = convT(dout).
extern "C" int ksr_chan_dw_bwd(const float* dout, const float* dw_w, float* din, int ld, int groups, int B, int H, int W,
                               hipStream_t stream) {
    KSR_CHECK_ARG(dout && dw_w && din, "chan_dw_bwd: null pointer");
    const bool st16 = ksr_prec() == KSR_BF16S;
    KSR_CHECK_ARG(groups >= 1 && groups <= 3 && ld >= 64 * groups && ld % (st16 ? 8 : 4) == 0, "chan_dw_bwd: bad channel layout");
    if (int rc = dims_ok(B, H, W, "chan_dw_bwd")) return rc;
    DwArgs a{dout, dw_w, nullptr, din, ld, groups, B, H, W};
    const long ntiles = (long)B * ((H + 3) / 4) * ((W + 31) / 32);
    const dim3 grid((unsigned)(ntiles < 1024 ? ntiles : 1024));
    if (st16) hipLaunchKernelGGL((chan_dw_kernel<__bf16, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((chan_dw_kernel<float, true>), grid, dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("chan_dw_bwd");
    return KSR_OK;
}

// ksr_chan_dw_bwd for the three branches at once (ld = 192, groups = 3) followed by the data gradient of their 1x1
// projections, in one kernel: din [N,192] = convT(dout), dx [N,64] = din W with pw_w = W [192][64] (the q | k | v projection
// weights).  bf16-storage mode only (every [N, C] tensor bf16).
extern "C" int ksr_chan_dw_bwd_dx(const float* dout, const float* dw_w, const float* pw_w, float* din, float* dx, int B, int H,
                                  int W, hipStream_t stream) {
    KSR_CHECK_ARG(dout && dw_w && pw_w && din && dx, "chan_dw_bwd_dx: null pointer");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "chan_dw_bwd_dx: bf16-storage mode only (ksr_chan_dw_bwd + ksr_pointwise_bwd otherwise)");
    if (int rc = dims_ok(B, H, W, "chan_dw_bwd_dx")) return rc;
    DwDxArgs a{dout, dw_w, pw_w, din, dx, B, H, W};
    const long ntiles = (long)B * ((H + 3) / 4) * ((W + 31) / 32);
    hipLaunchKernelGGL(chan_dw_bwd_dx_kernel, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("chan_dw_bwd_dx");
    return KSR_OK;
}

// From the per-image Gram matrix G = q^T k and the squared norms (tail = (sum q | sum q^2 | sum k^2), 192
// floats per image, as ksr_gram_batched emits with want_sq): A = softmax(q^^T k^), S = q^^T k^, raw norms,
// and M = W_a A (a_proj folded into the attention matrix).
extern "C" int ksr_chan_soft_fwd(const float* G, const float* tail, const float* a_proj_w, float* A, float* S,
                                 float* nrm, float* M, int n_out, int B, hipStream_t stream) {
    KSR_CHECK_ARG(G && tail && a_proj_w && A && S && nrm && M && n_out > 0 && B > 0, "chan_soft_fwd: bad argument");
    SoftArgs a{G, tail, a_proj_w, A, S, nrm, M, n_out};
    int ny = (n_out + 47) / 48;
    if (ny > 8) ny = 8;
    hipLaunchKernelGGL(chan_soft_fwd_kernel, dim3(B, ny), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("chan_soft_fwd");
    return KSR_OK;
}

// Backward of the per-image algebra.  Gd [B][n_out][64] = dy^T v.  Emits Mqk [B][128][128] with
// [dq | dk] = Mqk [q | k], and dW_a [n_out][64] (dWa_part: [B][n_out][64] workspace).
extern "C" int ksr_chan_soft_bwd(const float* Gd, const float* A, const float* S, const float* nrm,
                                 const float* a_proj_w, float* Mqk, float* dWa_part, float* dWa, int n_out, int B,
                                 hipStream_t stream) {
    KSR_CHECK_ARG(Gd && A && S && nrm && a_proj_w && Mqk && dWa_part && dWa && n_out > 0 && B > 0, "chan_soft_bwd: bad argument");
    SoftBwdArgs a{Gd, A, S, nrm, a_proj_w, Mqk, dWa_part, n_out};
    int K = (n_out + 63) / 64;
    if (K > 8) K = 8;
    hipLaunchKernelGGL(chan_soft_bwd_kernel, dim3(B, 1 + K), dim3(1024), 0, stream, a);
    KSR_LAUNCH_CHECK("chan_soft_bwd");
    const int n = n_out * 64;
    hipLaunchKernelGGL(sum_entries_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dWa_part, B, (long)n, n, dWa);
    KSR_LAUNCH_CHECK("chan_soft_bwd(sum)");
    return KSR_OK;
}

// out[i] = sum over B entries of in[b][i], i < n (entries `stride` floats apart): bias gradient from the
// per-image column sums.
extern "C" int ksr_sum_entries(const float* in, int B, int64_t stride, int n, float* out, hipStream_t stream) {
    KSR_CHECK_ARG(in && out && B > 0 && n > 0 && stride >= n, "sum_entries: bad argument");
    hipLaunchKernelGGL(sum_entries_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, B, (long)stride, n, out);
    KSR_LAUNCH_CHECK("sum_entries");
    return KSR_OK;
}

// Per-image 1x1 conv on NHWC pixels: out[n, out_c0 : out_c0+n_out] = Wm[b(n)] * in[n, in_c0 : in_c0+n_in] (+ bias),
// Wm [B][n_out][n_in] row-major, or [B][n_in][n_out] with transposed != 0.  Supported (n_in -> n_out):
// 64 -> {40, 64, 160, 336}; {40, 64, 160, 336} -> 64 ; 128 -> 128.
extern "C" int ksr_chan_apply(const float* in, int in_ld, int in_c0, int n_in, const float* Wm, int transposed,
                              const float* bias, float* out, int out_ld, int out_c0, int n_out, int B, int HW,
                              hipStream_t stream) {
    KSR_CHECK_ARG(in && Wm && out, "chan_apply: null pointer");
    KSR_CHECK_ARG(B > 0 && B <= 65535 && HW > 0 && (long)B * HW < (1L << 31), "chan_apply: bad dimensions");
    KSR_CHECK_ARG(n_in % 8 == 0 && n_out % 8 == 0 && in_ld % 4 == 0 && in_c0 % 4 == 0 && out_ld % 4 == 0 && out_c0 % 4 == 0,
                  "chan_apply: channel counts must be multiples of 8, offsets multiples of 4");
    ApArgs a{in, in_ld, in_c0, n_in / 2, Wm, (long)n_in * n_out, transposed ? n_out : n_in, bias,
             out, out_ld, out_c0, n_out / 2, HW};
    if (!transposed && n_in == 64) {
        if (n_out == 40 || n_out == 64) return launch_apply<32, 2, false>(a, B, stream);
        if (n_out == 160) return launch_apply<32, 5, false>(a, B, stream);
        if (n_out == 336) return launch_apply<32, 11, false>(a, B, stream);
    }
    if (!transposed && n_in == 128 && n_out == 128) return launch_apply<64, 4, false>(a, B, stream);
    if (transposed && n_out == 64) {
        if (n_in == 40) return launch_apply<24, 2, true>(a, B, stream);
        if (n_in == 64) return launch_apply<32, 2, true>(a, B, stream);
        if (n_in == 160) return launch_apply<80, 2, true>(a, B, stream);
        if (n_in == 336) return launch_apply<168, 2, true>(a, B, stream);
    }
    ksr_set_error("chan_apply: unsupported (n_in=%d, n_out=%d, transposed=%d)", n_in, n_out, transposed);
    return KSR_ERR_UNSUPPORTED;
}
