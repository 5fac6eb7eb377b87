// Stand-alone KAN-cell op (coefficient tensor materialised) and the chroma bilinear upsampler.
//
// kan_cell: per LR pixel n and output o
//   out[n,o] = sum_i rw[n,o,i]*silu(x[n,i]) + uw[n,o,i] * sum_g B_g(x[n,i]) * coef[n,o,i,g]
// with the per-pixel parameter vector w[n, :] = [coef (o,i,g) | uw (o,i) | rw (o,i)], optionally
// followed by the pixel shuffle out[b, 0, h*r+o/r, w*r+o%r].  HBM-bound: one thread per pixel,
// only the k+1 live coefficients per (o,i) are read.
#include "ksr_common.h"

namespace {

struct CellArgs {
    const float* x;   // [B, I, H, W]
    const float* w;   // NCHW [B, P, H, W] or NHWC [B*H*W, P]
    float* out;       // [B, O, H, W] or shuffled [B, O/r^2, rH, rW]
    const float* gout;  // backward: gradient of out (same layout as out)
    float* gw;        // backward: gradient of w (same layout as w)
    float* gx;        // backward: gradient of x (may be null)
    int B, I, O, H, W, Kc, P;
    int w_nhwc;
    int r;            // 0: no shuffle
    long out_bstride;  // elements between images in out (allows writing into a wider tensor)
    // kan_lin kernels: w[n, :] = lb + lw * ly[n, :] is computed on the fly instead of read
    const float* ly;  // [N, 4]
    const float* lw;  // [P][4]
    const float* lb;  // [P]
    float* dly;       // backward: gradient of ly [N, 4]
    float* lslab;     // backward: per-workgroup sums [gridDim.x][5][P] (rows 0..3: gradient of lw^T, row 4: of lb)
};

__device__ __forceinline__ long w_index(const CellArgs& a, int b, int hw, int p, int HW) {
    return a.w_nhwc ? ((long)(b * (long)HW + hw) * a.P + p) : (((long)b * a.P + p) * HW + hw);
}

__device__ __forceinline__ long out_index(const CellArgs& a, int b, int o, int h, int w) {
    if (a.r > 0) {
        const int rr = a.r * a.r;
        const int c = o / rr, q = o % rr;
        const int oy = h * a.r + q / a.r, ox = w * a.r + q % a.r;
        return (long)b * a.out_bstride + ((long)c * (a.H * a.r) + oy) * (long)(a.W * a.r) + ox;
    }
    return (long)b * a.out_bstride + ((long)o * a.H + h) * a.W + w;
}

template <int K, int OMAX>
__global__ void __launch_bounds__(256) kan_cell_fwd_kernel(CellArgs a, KsrKnots kn) {
    __shared__ float t[KSR_MAX_KNOTS];
    if (threadIdx.x < KSR_MAX_KNOTS) t[threadIdx.x] = kn.t[threadIdx.x < kn.M ? threadIdx.x : 0];
    __syncthreads();
    const int HW = a.H * a.W;
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (long)a.B * HW) return;
    const int b = (int)(n / HW), hw = (int)(n % HW);
    float acc[OMAX];
#pragma unroll
    for (int o = 0; o < OMAX; ++o) acc[o] = 0.f;
    const int uw_off = a.O * a.I * a.Kc, rw_off = uw_off + a.O * a.I;
    for (int i = 0; i < a.I; ++i) {
        const float x = a.x[((long)b * a.I + i) * HW + hw];
        float N[K + 1], dN[K + 1];
        const int m = ksr_bspline_local<K, false>(x, t, kn.M, N, dN);
        const float sl = ksr_silu(x);
#pragma unroll
        for (int o = 0; o < OMAX; ++o) {
            if (o < a.O) {
                float S = 0.f;
                if (m >= 0) {
#pragma unroll
                    for (int q = 0; q <= K; ++q) {
                        const int g = m - K + q;
                        if (g >= 0 && g < a.Kc) S += N[q] * a.w[w_index(a, b, hw, (o * a.I + i) * a.Kc + g, HW)];
                    }
                }
                const float uw = a.w[w_index(a, b, hw, uw_off + o * a.I + i, HW)];
                const float rw = a.w[w_index(a, b, hw, rw_off + o * a.I + i, HW)];
                acc[o] += rw * sl + uw * S;
            }
        }
    }
    const int h = hw / a.W, wx = hw % a.W;
#pragma unroll
    for (int o = 0; o < OMAX; ++o)
        if (o < a.O) a.out[out_index(a, b, o, h, wx)] = acc[o];
}

template <int K, int OMAX>
__global__ void __launch_bounds__(256) kan_cell_bwd_kernel(CellArgs a, KsrKnots kn) {
    __shared__ float t[KSR_MAX_KNOTS];
    if (threadIdx.x < KSR_MAX_KNOTS) t[threadIdx.x] = kn.t[threadIdx.x < kn.M ? threadIdx.x : 0];
    __syncthreads();
    const int HW = a.H * a.W;
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (long)a.B * HW) return;
    const int b = (int)(n / HW), hw = (int)(n % HW);
    const int h = hw / a.W, wx = hw % a.W;
    float go[OMAX];
#pragma unroll
    for (int o = 0; o < OMAX; ++o) go[o] = (o < a.O) ? a.gout[out_index(a, b, o, h, wx)] : 0.f;
    const int uw_off = a.O * a.I * a.Kc, rw_off = uw_off + a.O * a.I;
    for (int i = 0; i < a.I; ++i) {
        const float x = a.x[((long)b * a.I + i) * HW + hw];
        float N[K + 1], dN[K + 1];
        const int m = ksr_bspline_local<K, true>(x, t, kn.M, N, dN);
        const float sg = ksr_sigmoid(x);
        const float sl = x * sg;
        const float dsl = sg * (1.f + x * (1.f - sg));
        float gxi = 0.f;
#pragma unroll
        for (int o = 0; o < OMAX; ++o) {
            if (o < a.O) {
                const long iuw = w_index(a, b, hw, uw_off + o * a.I + i, HW);
                const long irw = w_index(a, b, hw, rw_off + o * a.I + i, HW);
                const float uw = a.w[iuw], rw = a.w[irw];
                float S = 0.f, dS = 0.f;
                for (int g = 0; g < a.Kc; ++g) {
                    const int q = g - (m - K);
                    float bg = 0.f, dbg = 0.f;
                    if (m >= 0) {
#pragma unroll
                        for (int qq = 0; qq <= K; ++qq)
                            if (qq == q) { bg = N[qq]; dbg = dN[qq]; }
                    }
                    const long ic = w_index(a, b, hw, (o * a.I + i) * a.Kc + g, HW);
                    if (bg != 0.f || dbg != 0.f) {
                        const float c = a.w[ic];
                        S += bg * c;
                        dS += dbg * c;
                    }
                    a.gw[ic] = go[o] * uw * bg;
                }
                a.gw[iuw] = go[o] * S;
                a.gw[irw] = go[o] * sl;
                gxi += go[o] * (rw * dsl + uw * dS);
            }
        }
        if (a.gx) a.gx[((long)b * a.I + i) * HW + hw] = gxi;
    }
}

// ------------------------------------------------------------------------------------------ NHWC fast path
// w [N, P]: the P parameters of a pixel are contiguous, so one-pixel-per-thread accesses are strided by P.  These
// kernels stage the [32 pixels][P] block of a workgroup (one contiguous range of w) in LDS with lane-linear 16-byte
// loads, let 8 threads share a pixel (thread (px, sub) owns outputs o = sub, sub + 8, ...), and in the backward build
// the gradient block in place in LDS and write it back with lane-linear stores.
constexpr int NH_TP = 32;          // pixels per workgroup
constexpr int NH_PAD = 4;          // LDS row padding (floats)

template <int K, bool BWD>
__global__ void __launch_bounds__(256) kan_cell_nhwc_kernel(CellArgs a, KsrKnots kn) {
    extern __shared__ __attribute__((aligned(16))) float wt[];      // [NH_TP][P + NH_PAD]
    __shared__ float t[KSR_MAX_KNOTS];
    const int tid = threadIdx.x;
    if (tid < KSR_MAX_KNOTS) t[tid] = kn.t[tid < kn.M ? tid : 0];
    const int HW = a.H * a.W;
    const long N = (long)a.B * HW;
    const long n0 = (long)blockIdx.x * NH_TP;
    const int npx = (N - n0) < NH_TP ? (int)(N - n0) : NH_TP;
    const int P = a.P, LD = P + NH_PAD, P4 = P / 4;
    // stage the block (P is a multiple of 4: checked by the host)
    const float4* src = reinterpret_cast<const float4*>(a.w + n0 * P);
    for (int i = tid; i < npx * P4; i += 256) {
        const int r = i / P4, q = i - r * P4;
        *reinterpret_cast<float4*>(wt + r * LD + 4 * q) = src[i];
    }
    __syncthreads();
    const int px = tid >> 3, sub = tid & 7;
    const long n = n0 + px;
    const bool live = px < npx;
    const int b = live ? (int)(n / HW) : 0, hw = live ? (int)(n % HW) : 0;
    const int h = hw / a.W, wx = hw % a.W;
    float* row = wt + px * LD;
    const int uw_off = a.O * a.I * a.Kc, rw_off = uw_off + a.O * a.I;
    for (int i = 0; i < a.I; ++i) {
        const float x = live ? a.x[((long)b * a.I + i) * HW + hw] : 0.f;
        float Nv[K + 1], dN[K + 1];
        const int m = ksr_bspline_local<K, BWD>(x, t, kn.M, Nv, dN);
        const float sg = ksr_sigmoid(x);
        const float sl = x * sg;
        const float dsl = sg * (1.f + x * (1.f - sg));
        float gxi = 0.f;
        for (int o = sub; o < a.O; o += 8) {
            float* cf = row + (o * a.I + i) * a.Kc;
            const float uw = row[uw_off + o * a.I + i], rw = row[rw_off + o * a.I + i];
            float S = 0.f, dS = 0.f;
            if (m >= 0) {
#pragma unroll
                for (int q = 0; q <= K; ++q) {
                    const int g = m - K + q;
                    if (g >= 0 && g < a.Kc) { S += Nv[q] * cf[g]; if (BWD) dS += dN[q] * cf[g]; }
                }
            }
            if (!BWD) {
                if (live) {
                    // accumulate over i in the output (I > 1 is rare: read-modify-write of this thread's own element)
                    const long oi = out_index(a, b, o, h, wx);
                    const float v = rw * sl + uw * S;
                    a.out[oi] = (i == 0) ? v : a.out[oi] + v;
                }
            } else {
                const float go = live ? a.gout[out_index(a, b, o, h, wx)] : 0.f;
                for (int g = 0; g < a.Kc; ++g) {
                    float bg = 0.f;
                    if (m >= 0) {
#pragma unroll
                        for (int q = 0; q <= K; ++q) if (g == m - K + q) bg = Nv[q];
                    }
                    cf[g] = go * uw * bg;
                }
                row[uw_off + o * a.I + i] = go * S;
                row[rw_off + o * a.I + i] = go * sl;
                gxi += go * (rw * dsl + uw * dS);
            }
        }
        if (BWD && a.gx) {
            gxi += __shfl_xor(gxi, 1, 64); gxi += __shfl_xor(gxi, 2, 64); gxi += __shfl_xor(gxi, 4, 64);
            if (live && sub == 0) a.gx[((long)b * a.I + i) * HW + hw] = gxi;
        }
    }
    if (BWD) {
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(a.gw + n0 * P);
        for (int i = tid; i < npx * P4; i += 256) {
            const int r = i / P4, q = i - r * P4;
            dst[i] = *reinterpret_cast<const float4*>(wt + r * LD + 4 * q);
        }
    }
}

template <int K>
int launch_cell_nhwc(bool bwd, const CellArgs& a, const KsrKnots& kn, hipStream_t s) {
    const long n = (long)a.B * a.H * a.W;
    const size_t lds = (size_t)NH_TP * (a.P + NH_PAD) * sizeof(float);
    const dim3 grid((unsigned)((n + NH_TP - 1) / NH_TP)), block(256);
    if (bwd) {
        static bool set_b = false;
        if (!set_b) { (void)hipFuncSetAttribute((const void*)kan_cell_nhwc_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); set_b = true; }
        hipLaunchKernelGGL((kan_cell_nhwc_kernel<K, true>), grid, block, lds, s, a, kn);
    } else {
        static bool set_f = false;
        if (!set_f) { (void)hipFuncSetAttribute((const void*)kan_cell_nhwc_kernel<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); set_f = true; }
        hipLaunchKernelGGL((kan_cell_nhwc_kernel<K, false>), grid, block, lds, s, a, kn);
    }
    KSR_LAUNCH_CHECK("kan_cell(nhwc)");
    return KSR_OK;
}

// ------------------------------------------------------------------------------------------ 1x1 conv + KAN cell, fused
// The UNet hyper-encoder ends in a 1x1 conv 4 -> P (OutConv, conv_sep_kan/unet.py:28,41) whose output is the KAN cell's
// per-pixel parameter vector: P = 160 floats per pixel written, read back, and in the backward read and written once
// more, re-read by the conv's data gradient and by its weight gradient.  These kernels build the [32 pixels][P] block of
// a workgroup in LDS from the 4 input channels (w = lb + lw y), run the NHWC cell code above on it and, in the backward,
// consume the gradient block in place: dy = gw lw (8 threads per pixel, combined by shuffles), and the sums over pixels
// gw^T y and gw^T 1 in registers of the thread that owns entry p (tid, tid + 256), carried across the tiles of a
// persistent workgroup and written once as a partial row.  I = 1 only.
template <int K, bool BWD>
__global__ void __launch_bounds__(256) kan_lin_kernel(CellArgs a, KsrKnots kn) {
    extern __shared__ __attribute__((aligned(16))) float wt[];      // [NH_TP][P + NH_PAD] | lw [P][4] | lb [P] | y [NH_TP][4]
    __shared__ float t[KSR_MAX_KNOTS];
    const int tid = threadIdx.x;
    const int P = a.P, LD = P + NH_PAD;
    float4* lw4 = reinterpret_cast<float4*>(wt + NH_TP * LD);
    float* lbs = wt + NH_TP * LD + 4 * P;
    float4* ys = reinterpret_cast<float4*>(lbs + P);
    if (tid < KSR_MAX_KNOTS) t[tid] = kn.t[tid < kn.M ? tid : 0];
    for (int i = tid; i < 4 * P; i += 256) reinterpret_cast<float*>(lw4)[i] = a.lw[i];      // parameters may sit at any float offset
    for (int p = tid; p < P; p += 256) lbs[p] = a.lb[p];
    const int HW = a.H * a.W;
    const long N = (long)a.B * HW;
    const long ntiles = (N + NH_TP - 1) / NH_TP;
    const int px = tid >> 3, sub = tid & 7;
    const int uw_off = a.O * a.Kc, rw_off = uw_off + a.O;
    float accw[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accb[2] = {0.f, 0.f};
    const int r_first = tid / P, p_first = tid - r_first * P;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long n0 = tile * NH_TP;
        const int npx = (N - n0) < NH_TP ? (int)(N - n0) : NH_TP;
        // every global load of the tile is issued here, ahead of the block build that hides its latency
        const long n = n0 + px;
        const bool live = px < npx;
        const int b = live ? (int)(n / HW) : 0, hw = live ? (int)(n % HW) : 0;
        const int h = hw / a.W, wx = hw % a.W;
        const float x = live ? a.x[(long)b * HW + hw] : 0.f;
        float gop[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gop[j] = (BWD && live && sub + 8 * j < a.O) ? a.gout[out_index(a, b, sub + 8 * j, h, wx)] : 0.f;
        const float4 ynew = (tid < npx) ? reinterpret_cast<const float4*>(a.ly)[n0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();                                   // the previous tile's block is no longer read
        if (tid < NH_TP) ys[tid] = ynew;
        __syncthreads();
        for (int r = r_first, p = p_first; r < NH_TP; ) {          // element (r, p), 256 elements further per step
            const float4 y = ys[r], w4 = lw4[p];
            wt[r * LD + p] = fmaf(w4.w, y.w, fmaf(w4.z, y.z, fmaf(w4.y, y.y, fmaf(w4.x, y.x, lbs[p]))));
            p += 256;
            while (p >= P) { p -= P; ++r; }
        }
        __syncthreads();
        float* row = wt + px * LD;
        float Nv[K + 1], dN[K + 1];
        const int m = ksr_bspline_local<K, false>(x, t, kn.M, Nv, dN);
        const float sl = ksr_silu(x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = sub + 8 * j;
            if (o >= a.O) break;
            float* cf = row + o * a.Kc;
            const float uw = row[uw_off + o], rw = row[rw_off + o];
            float S = 0.f;
            if (m >= 0) {
#pragma unroll
                for (int q = 0; q <= K; ++q) {
                    const int g = m - K + q;
                    if (g >= 0 && g < a.Kc) S += Nv[q] * cf[g];
                }
            }
            if (!BWD) {
                if (live) a.out[out_index(a, b, o, h, wx)] = rw * sl + uw * S;
            } else {
                const float go = gop[j];
                for (int g = 0; g < a.Kc; ++g) {
                    float bg = 0.f;
                    if (m >= 0) {
#pragma unroll
                        for (int q = 0; q <= K; ++q) if (g == m - K + q) bg = Nv[q];
                    }
                    cf[g] = go * uw * bg;
                }
                row[uw_off + o] = go * S;
                row[rw_off + o] = go * sl;
            }
        }
        if (BWD) {
            __syncthreads();
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
            for (int p = sub; p < P; p += 8) {
                const float g = row[p];
                const float4 w4 = lw4[p];
                d0 = fmaf(g, w4.x, d0); d1 = fmaf(g, w4.y, d1); d2 = fmaf(g, w4.z, d2); d3 = fmaf(g, w4.w, d3);
            }
#pragma unroll
            for (int sh = 1; sh < 8; sh <<= 1) {
                d0 += __shfl_xor(d0, sh, 64); d1 += __shfl_xor(d1, sh, 64); d2 += __shfl_xor(d2, sh, 64); d3 += __shfl_xor(d3, sh, 64);
            }
            if (live && sub == 0) reinterpret_cast<float4*>(a.dly)[n] = make_float4(d0, d1, d2, d3);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int p = tid + 256 * q;
                if (p < P) {
                    for (int r = 0; r < npx; ++r) {
                        const float g = wt[r * LD + p];
                        const float4 y = ys[r];
                        accw[q][0] = fmaf(g, y.x, accw[q][0]); accw[q][1] = fmaf(g, y.y, accw[q][1]);
                        accw[q][2] = fmaf(g, y.z, accw[q][2]); accw[q][3] = fmaf(g, y.w, accw[q][3]);
                        accb[q] += g;
                    }
                }
            }
        }
    }
    if (BWD) {
        float* s = a.lslab + (long)blockIdx.x * 5 * P;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q;
            if (p < P) {
#pragma unroll
                for (int c = 0; c < 4; ++c) s[c * P + p] = accw[q][c];
                s[4 * P + p] = accb[q];
            }
        }
    }
}

// Forward: one pixel per lane, no parameter block at all: only the 4 live coefficients, uw and rw of each output are
// evaluated (6 dot products of length 4 per output), with the same fused-multiply-add order as the block form above.
template <int K>
__global__ void __launch_bounds__(256) kan_lin_fwd_kernel(CellArgs a, KsrKnots kn) {
    extern __shared__ __attribute__((aligned(16))) float wt[];      // lw [P][4] | lb [P]
    __shared__ float t[KSR_MAX_KNOTS];
    const int tid = threadIdx.x, P = a.P;
    const float4* lw4 = reinterpret_cast<const float4*>(wt);
    const float* lbs = wt + 4 * P;
    if (tid < KSR_MAX_KNOTS) t[tid] = kn.t[tid < kn.M ? tid : 0];
    for (int i = tid; i < 4 * P; i += 256) wt[i] = a.lw[i];
    for (int p = tid; p < P; p += 256) wt[4 * P + p] = a.lb[p];
    __syncthreads();
    const int HW = a.H * a.W;
    const long n = (long)blockIdx.x * 256 + tid;
    if (n >= (long)a.B * HW) return;
    const int b = (int)(n / HW), hw = (int)(n % HW);
    const int h = hw / a.W, wx = hw % a.W;
    const float4 y = reinterpret_cast<const float4*>(a.ly)[n];
    const float x = a.x[n];
    float Nv[K + 1], dN[K + 1];
    const int m = ksr_bspline_local<K, false>(x, t, kn.M, Nv, dN);
    const float sl = ksr_silu(x);
    const int uw_off = a.O * a.Kc, rw_off = uw_off + a.O;
    auto param = [&](int p) {
        const float4 w4 = lw4[p];
        return fmaf(w4.w, y.w, fmaf(w4.z, y.z, fmaf(w4.y, y.y, fmaf(w4.x, y.x, lbs[p]))));
    };
    for (int o0 = 0; o0 < a.O; o0 += 4) {
        float res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + j;
            res[j] = 0.f;
            if (o < a.O) {
                float S = 0.f;
                if (m >= 0) {
#pragma unroll
                    for (int q = 0; q <= K; ++q) {
                        const int g = m - K + q;
                        if (g >= 0 && g < a.Kc) S += Nv[q] * param(o * a.Kc + g);
                    }
                }
                res[j] = param(rw_off + o) * sl + param(uw_off + o) * S;
            }
        }
        if (a.r == 4 && o0 + 4 <= a.O)        // the 4 outputs are 4 neighbouring pixels of one output row
            *reinterpret_cast<float4*>(a.out + out_index(a, b, o0, h, wx)) = make_float4(res[0], res[1], res[2], res[3]);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (o0 + j < a.O) a.out[out_index(a, b, o0 + j, h, wx)] = res[j];
    }
}

// Backward for O = 16 outputs (upscale 4) and 4 (Kc + 2) <= 64: one pixel per lane, a wave owns 64 pixels and works through
// the outputs in 4 chunks of 4.  Per chunk a lane evaluates its pixel's live parameters (as the forward), adds their
// gradients' share to dy in registers, and writes the chunk's 4 (Kc + 2) gradient entries (zeros for the coefficients
// outside the pixel's knot interval) to its row of a wave-private LDS tile [64][4 (Kc + 2) + 1]; then lane e sums entry e
// over the 64 rows against y (gw^T y, gw^T 1) into 5 accumulators per chunk that it keeps across all its tiles.  No
// workgroup barrier inside the loop: the tile is private to the wave.  About a third of the instructions per pixel of
// the 8-threads-per-pixel kernel above (which evaluates the basis 8 times per pixel and builds all P parameters).
template <int K>
__global__ void __launch_bounds__(256) kan_lin_bwd16_kernel(CellArgs a, KsrKnots kn) {
    extern __shared__ __attribute__((aligned(16))) float wt[];      // lw [P][4] | lb [P] | per wave: ys [64][4], g [64][EC + 1]
    __shared__ float t[KSR_MAX_KNOTS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int P = a.P, Kc = a.Kc, EC = 4 * (Kc + 2), LDG = EC + 1;
    const float4* lw4 = reinterpret_cast<const float4*>(wt);
    const float* lbs = wt + 4 * P;
    float4* ysw = reinterpret_cast<float4*>(wt + 5 * P + wv * (256 + 64 * LDG));
    float* gt = reinterpret_cast<float*>(ysw) + 256;
    if (tid < KSR_MAX_KNOTS) t[tid] = kn.t[tid < kn.M ? tid : 0];
    for (int i = tid; i < 4 * P; i += 256) wt[i] = a.lw[i];
    for (int p = tid; p < P; p += 256) wt[4 * P + p] = a.lb[p];
    __syncthreads();
    const int HW = a.H * a.W;
    const long N = (long)a.B * HW;
    const long ntiles = (N + 63) / 64;
    const int uw_off = 16 * Kc, rw_off = uw_off + 16;
    float acc[4][5];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[c][i] = 0.f;
    float* myrow = gt + lane * LDG;
    for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long n = tile * 64 + lane;
        const bool live = n < N;
        const int b = live ? (int)(n / HW) : 0, hw = live ? (int)(n % HW) : 0;
        const int h = hw / a.W, wx = hw % a.W;
        const float4 y = live ? reinterpret_cast<const float4*>(a.ly)[n] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float x = live ? a.x[n] : 0.f;
        float go[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {           // outputs 4i..4i+3 are 4 neighbouring pixels of output row 4h + i
            const float4 v = live ? *reinterpret_cast<const float4*>(a.gout + out_index(a, b, 4 * i, h, wx)) : make_float4(0.f, 0.f, 0.f, 0.f);
            go[4 * i] = v.x; go[4 * i + 1] = v.y; go[4 * i + 2] = v.z; go[4 * i + 3] = v.w;
        }
        float Nv[K + 1], dN[K + 1];
        const int m = ksr_bspline_local<K, false>(x, t, kn.M, Nv, dN);
        const float sl = ksr_silu(x);
        auto param = [&](int p) {
            const float4 w4 = lw4[p];
            return fmaf(w4.w, y.w, fmaf(w4.z, y.z, fmaf(w4.y, y.y, fmaf(w4.x, y.x, lbs[p]))));
        };
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        auto emit = [&](int e, int p, float g) {           // gradient g of parameter p: tile entry e, and its share of dy
            myrow[e] = g;
            const float4 w4 = lw4[p];
            d0 = fmaf(g, w4.x, d0); d1 = fmaf(g, w4.y, d1); d2 = fmaf(g, w4.z, d2); d3 = fmaf(g, w4.w, d3);
        };
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                    // the previous tile's rows and ys are no longer read
        ysw[lane] = y;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            for (int e = 0; e < 4 * Kc; ++e) myrow[e] = 0.f;
#pragma unroll
            for (int jo = 0; jo < 4; ++jo) {
                const int o = 4 * c + jo;
                const float uw = param(uw_off + o);
                float S = 0.f;
                if (m >= 0) {
#pragma unroll
                    for (int q = 0; q <= K; ++q) {
                        const int g = m - K + q;
                        if (g >= 0 && g < Kc) {
                            S += Nv[q] * param(o * Kc + g);
                            emit(jo * Kc + g, o * Kc + g, go[o] * uw * Nv[q]);
                        }
                    }
                }
                emit(4 * Kc + jo, uw_off + o, go[o] * S);
                emit(4 * Kc + 4 + jo, rw_off + o, go[o] * sl);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                // all 64 rows of the chunk are written
            if (lane < EC) {
#pragma unroll 4
                for (int r = 0; r < 64; ++r) {
                    const float g = gt[r * LDG + lane];
                    const float4 yr = ysw[r];
                    acc[c][0] = fmaf(g, yr.x, acc[c][0]); acc[c][1] = fmaf(g, yr.y, acc[c][1]);
                    acc[c][2] = fmaf(g, yr.z, acc[c][2]); acc[c][3] = fmaf(g, yr.w, acc[c][3]);
                    acc[c][4] += g;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                // the rows are free for the next chunk
        }
        if (live) reinterpret_cast<float4*>(a.dly)[n] = make_float4(d0, d1, d2, d3);
    }
    if (lane < EC) {
        float* s = a.lslab + ((long)blockIdx.x * 4 + wv) * 5 * P;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = lane < 4 * Kc ? 4 * c * Kc + lane : (lane < 4 * Kc + 4 ? uw_off + 4 * c + (lane - 4 * Kc) : rw_off + 4 * c + (lane - 4 * Kc - 4));
#pragma unroll
            for (int i = 0; i < 5; ++i) s[i * P + p] = acc[c][i];
        }
    }
}

// out[e] = sum over the rows of slab [nrows][E]; 16 entries x 16 row slices per workgroup (a slice takes every 16th row, 8
// loads in flight), slices combined in fixed order
__global__ void __launch_bounds__(256) kan_lin_reduce_kernel(const float* __restrict__ slab, int nrows, int E, float* __restrict__ out) {
    __shared__ float part[16][17];
    const int le = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + le;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < E) {
        int r = sl;
        for (; r + 7 * 16 < nrows; r += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += slab[(long)(r + 16 * u) * E + e];
        }
        for (; r < nrows; r += 16) acc[0] += slab[(long)r * E + e];
    }
    part[sl][le] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (sl == 0 && e < E) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += part[q][le];
        out[e] = v;
    }
}

constexpr int LIN_MAX_WG = 1536;          // 6 workgroups of 25 KB LDS per CU: all resident at once
constexpr int LIN_MAX_P = 512;

int lin_grid(int B, int H, int W) {
    const long ntiles = ((long)B * H * W + NH_TP - 1) / NH_TP;
    return (int)(ntiles < LIN_MAX_WG ? ntiles : LIN_MAX_WG);
}

constexpr int LIN16_MAX_WG = 768;          // 3 workgroups of ~50 KB LDS per CU
int lin16_grid(int B, int H, int W) {
    const long nwg = ((long)B * H * W + 255) / 256;
    return (int)(nwg < LIN16_MAX_WG ? nwg : LIN16_MAX_WG);
}
bool lin16_fits(const CellArgs& a) {       // the one-pixel-per-lane backward: 16 outputs pixel-shuffled by 4, aligned gradient rows
    return a.O == 16 && a.r == 4 && 4 * (a.Kc + 2) <= 64 && (reinterpret_cast<uintptr_t>(a.gout) & 15) == 0 && a.out_bstride % 4 == 0;
}
// partial rows the backward writes to the slab
int lin_slab_rows(bool fits16, int B, int H, int W) { return fits16 ? 4 * lin16_grid(B, H, W) : lin_grid(B, H, W); }

template <int K>
int launch_lin(bool bwd, const CellArgs& a, const KsrKnots& kn, hipStream_t s) {
    const size_t lds = ((size_t)NH_TP * (a.P + NH_PAD) + 5 * (size_t)a.P + 4 * NH_TP) * sizeof(float);
    const dim3 grid((unsigned)lin_grid(a.B, a.H, a.W)), block(256);
    if (bwd && lin16_fits(a)) {
        const size_t lds16 = (5 * (size_t)a.P + 4 * (256 + 64 * (size_t)(4 * (a.Kc + 2) + 1))) * sizeof(float);
        static bool set_6 = false;
        if (!set_6) { (void)hipFuncSetAttribute((const void*)kan_lin_bwd16_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); set_6 = true; }
        hipLaunchKernelGGL((kan_lin_bwd16_kernel<K>), dim3((unsigned)lin16_grid(a.B, a.H, a.W)), block, lds16, s, a, kn);
    } else if (bwd) {
        static bool set_b = false;
        if (!set_b) { (void)hipFuncSetAttribute((const void*)kan_lin_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); set_b = true; }
        hipLaunchKernelGGL((kan_lin_kernel<K, true>), grid, block, lds, s, a, kn);
    } else {
        const long n = (long)a.B * a.H * a.W;
        hipLaunchKernelGGL((kan_lin_fwd_kernel<K>), dim3((unsigned)((n + 255) / 256)), block, 5 * (size_t)a.P * sizeof(float), s, a, kn);
    }
    KSR_LAUNCH_CHECK("kan_lin");
    return KSR_OK;
}

// ------------------------------------------------------------------------------------------ bilinear
// out[b, oc0 + c, :, :] = bilinear_up(in[b, ic0 + c, :, :], r), c < nc.  align_corners = false.
struct BilArgs {
    const float* in;
    float* out;
    int B, nc, H, W, r;
    long in_bstride, in_cstride, out_bstride, out_cstride;
    int ic0, oc0;
};

__global__ void __launch_bounds__(256) bilinear_up_kernel(BilArgs a) {
    const int OW = a.W * a.r, OH = a.H * a.r;
    const int OW4 = OW / 4;  // r in {2,4,...}: OW % 4 == 0 is checked by the host
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.nc * OH * OW4;
    if (idx >= total) return;
    const int x4 = (int)(idx % OW4);
    const int oy = (int)((idx / OW4) % OH);
    const int c = (int)((idx / ((long)OW4 * OH)) % a.nc);
    const int b = (int)(idx / ((long)OW4 * OH * a.nc));
    const float sc = 1.0f / (float)a.r;
    float sy = sc * ((float)oy + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
    const float* p = a.in + (long)b * a.in_bstride + (long)(a.ic0 + c) * a.in_cstride;
    const float* r0 = p + (long)y0 * a.W;
    const float* r1 = p + (long)y1 * a.W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ox = x4 * 4 + e;
        float sx = sc * ((float)ox + 0.5f) - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        const int x0 = (int)sx;
        const int x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
        const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
        v[e] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
    }
    float* o = a.out + (long)b * a.out_bstride + (long)(a.oc0 + c) * a.out_cstride + (long)oy * OW + x4 * 4;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
}

// The same values from a (64, 4) workgroup on a (column blocks, row blocks, image x plane) grid: a thread owns 4 output
// columns of the R output rows of one source row (no index divisions; the R float4 stores of a wave row are 1 KB
// contiguous each).  The expression per output is the one above, term for term.
template <int R>
__global__ void __launch_bounds__(256) bilinear_up_rows_kernel(BilArgs a) {
    const int OW = a.W * R, OW4 = OW / 4;
    const int x4 = blockIdx.x * 64 + threadIdx.x, ys = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= OW4 || ys >= a.H) return;
    const int b = blockIdx.z / a.nc, c = blockIdx.z - b * a.nc;
    const float sc = 1.0f / (float)R;
    const float* p = a.in + (long)b * a.in_bstride + (long)(a.ic0 + c) * a.in_cstride;
    int x0[4], x1[4];
    float lx0[4], lx1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ox = x4 * 4 + e;
        float sx = sc * ((float)ox + 0.5f) - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        x0[e] = (int)sx;
        x1[e] = x0[e] + (x0[e] < a.W - 1 ? 1 : 0);
        lx1[e] = sx - (float)x0[e];
        lx0[e] = 1.f - lx1[e];
    }
    float* o = a.out + (long)b * a.out_bstride + (long)(a.oc0 + c) * a.out_cstride + (long)ys * R * OW + x4 * 4;
#pragma unroll
    for (int ry = 0; ry < R; ++ry) {
        const int oy = ys * R + ry;
        float sy = sc * ((float)oy + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        const float* r0 = p + (long)y0 * a.W;
        const float* r1 = p + (long)y1 * a.W;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = ly0 * (lx0[e] * r0[x0[e]] + lx1[e] * r0[x1[e]]) + ly1 * (lx0[e] * r1[x0[e]] + lx1[e] * r1[x1[e]]);
        *reinterpret_cast<float4*>(o + (long)ry * OW) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Adjoint of bilinear_up_kernel: a.in = gradient of the upsampled planes, a.out = gradient of the source planes
// (in / out keep the forward's roles in BilArgs: `out` strides describe the r x larger tensor).  Gather form: every
// source pixel visits the upsampled pixels whose footprint can contain it and re-derives their weights.
__global__ void __launch_bounds__(256) bilinear_up_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, BilArgs a) {
    const int OW = a.W * a.r, OH = a.H * a.r;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.nc * a.H * a.W;
    if (idx >= total) return;
    const int x = (int)(idx % a.W);
    const int y = (int)((idx / a.W) % a.H);
    const int c = (int)((idx / ((long)a.W * a.H)) % a.nc);
    const int b = (int)(idx / ((long)a.W * a.H * a.nc));
    const float sc = 1.0f / (float)a.r;
    const float* g = gout + (long)b * a.out_bstride + (long)(a.oc0 + c) * a.out_cstride;
    const int oy_lo = max(0, a.r * (y - 1)), oy_hi = min(OH - 1, a.r * (y + 2) - 1);
    const int ox_lo = max(0, a.r * (x - 1)), ox_hi = min(OW - 1, a.r * (x + 2) - 1);
    float s = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        float sy = sc * ((float)oy + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            float sx = sc * ((float)ox + 0.5f) - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            const int x0 = (int)sx;
            const int x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
            const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
            s += wy * wx * g[(long)oy * OW + ox];
        }
    }
    gin[(long)b * a.in_bstride + (long)(a.ic0 + c) * a.in_cstride + (long)y * a.W + x] = s;
}

// Gradient of the first FFN's front (u = w1 Y + b1, pre = bd + dw3x3(u)) with respect to the luma plane:
// gx[n] += w1 * sum_tap wd[tap] * dpre(n - tap)   (taps that leave the image contribute nothing)
__global__ void __launch_bounds__(256) stem_dx_kernel(const float* __restrict__ dpre, const float* __restrict__ p1w,
                                                      const float* __restrict__ dww, float* __restrict__ gx, long gx_bstride,
                                                      int B, int H, int W) {
    const long HW = (long)H * W;
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (long)B * HW) return;
    const int b = (int)(n / HW), hw = (int)(n % HW);
    const int y = hw / W, x = hw % W;
    const float* dp = dpre + (long)b * HW;
    float du = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ys = y - dy, xs = x - dx;
            if (ys >= 0 && ys < H && xs >= 0 && xs < W) du += dww[(dy + 1) * 3 + (dx + 1)] * dp[(long)ys * W + xs];
        }
    gx[(long)b * gx_bstride + hw] += p1w[0] * du;
}

__global__ void __launch_bounds__(256) bilinear_up_scalar_kernel(BilArgs a) {
    const int OW = a.W * a.r, OH = a.H * a.r;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.nc * OH * OW;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int c = (int)((idx / ((long)OW * OH)) % a.nc);
    const int b = (int)(idx / ((long)OW * OH * a.nc));
    const float sc = 1.0f / (float)a.r;
    float sy = sc * ((float)oy + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    float sx = sc * ((float)ox + 0.5f) - 0.5f;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
    const float* p = a.in + (long)b * a.in_bstride + (long)(a.ic0 + c) * a.in_cstride;
    a.out[(long)b * a.out_bstride + (long)(a.oc0 + c) * a.out_cstride + (long)oy * OW + ox] =
        ly0 * (lx0 * p[(long)y0 * a.W + x0] + lx1 * p[(long)y0 * a.W + x1]) +
        ly1 * (lx0 * p[(long)y1 * a.W + x0] + lx1 * p[(long)y1 * a.W + x1]);
}

template <int K>
int launch_cell(bool bwd, const CellArgs& a, const KsrKnots& kn, hipStream_t s) {
    const long n = (long)a.B * a.H * a.W;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define KSR_CELL_DISPATCH(OM)                                                        \
    do {                                                                             \
        if (bwd) hipLaunchKernelGGL((kan_cell_bwd_kernel<K, OM>), grid, block, 0, s, a, kn); \
        else hipLaunchKernelGGL((kan_cell_fwd_kernel<K, OM>), grid, block, 0, s, a, kn);     \
    } while (0)
    if (a.O <= 4) KSR_CELL_DISPATCH(4);
    else if (a.O <= 16) KSR_CELL_DISPATCH(16);
    else KSR_CELL_DISPATCH(64);
#undef KSR_CELL_DISPATCH
    KSR_LAUNCH_CHECK("kan_cell");
    return KSR_OK;
}

int fill_knots(KsrKnots& kn, const float* knots, int n_knots, int k, int Kc) {
    KSR_CHECK_ARG(knots != nullptr, "knots is null");
    KSR_UNSUPPORTED(k < 0 || k > KSR_MAX_ORDER, "spline_order %d not supported (0..%d)", k, KSR_MAX_ORDER);
    KSR_UNSUPPORTED(n_knots > KSR_MAX_KNOTS, "%d knots exceed the limit of %d", n_knots, KSR_MAX_KNOTS);
    KSR_CHECK_ARG(n_knots == Kc + k + 1, "n_knots (%d) must equal grid_size + 2*spline_order + 1 = %d", n_knots, Kc + k + 1);
    for (int i = 0; i < KSR_MAX_KNOTS; ++i) kn.t[i] = i < n_knots ? knots[i] : knots[n_knots - 1];
    kn.M = n_knots;
    kn.k = k;
    return KSR_OK;
}

int cell_common(bool bwd, CellArgs& a, const float* knots, int n_knots, int k, hipStream_t s) {
    KSR_CHECK_ARG(a.B > 0 && a.I > 0 && a.O > 0 && a.H > 0 && a.W > 0 && a.Kc > 0, "kan_cell: non-positive dimension");
    KSR_UNSUPPORTED(a.O > 64, "kan_cell: out_dim %d > 64 not supported", a.O);
    KSR_CHECK_ARG(a.r == 0 || a.O % (a.r * a.r) == 0, "kan_cell: out_dim %d not divisible by r^2 (r=%d)", a.O, a.r);
    a.P = a.O * a.I * (a.Kc + 2);
    KsrKnots kn;
    int rc = fill_knots(kn, knots, n_knots, k, a.Kc);
    if (rc) return rc;
    // NHWC parameter tensors go through the LDS-staged kernels when a 32-pixel block fits (P <= ~980)
    const bool staged = a.w_nhwc && a.P % 4 == 0 && (size_t)NH_TP * (a.P + NH_PAD) * sizeof(float) <= 128 * 1024 &&
                        (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (!bwd || (reinterpret_cast<uintptr_t>(a.gw) & 15) == 0);
    if (staged) {
        switch (k) {
            case 0: return launch_cell_nhwc<0>(bwd, a, kn, s);
            case 1: return launch_cell_nhwc<1>(bwd, a, kn, s);
            case 2: return launch_cell_nhwc<2>(bwd, a, kn, s);
            default: return launch_cell_nhwc<3>(bwd, a, kn, s);
        }
    }
    switch (k) {
        case 0: return launch_cell<0>(bwd, a, kn, s);
        case 1: return launch_cell<1>(bwd, a, kn, s);
        case 2: return launch_cell<2>(bwd, a, kn, s);
        default: return launch_cell<3>(bwd, a, kn, s);
    }
}

}  // namespace

extern "C" int ksr_kan_cell_fwd(const float* x, const float* w, float* out, int B, int I, int O, int H, int W,
                                int n_coef, const float* knots, int n_knots, int spline_order, int w_nhwc,
                                int shuffle_r, int64_t out_batch_stride, hipStream_t stream) {
    KSR_CHECK_ARG(x && w && out, "kan_cell_fwd: null pointer");
    CellArgs a{};
    a.x = x; a.w = w; a.out = out;
    a.B = B; a.I = I; a.O = O; a.H = H; a.W = W; a.Kc = n_coef; a.w_nhwc = w_nhwc; a.r = shuffle_r;
    a.out_bstride = out_batch_stride > 0 ? out_batch_stride : (long)O * H * W;
    return cell_common(false, a, knots, n_knots, spline_order, stream);
}

namespace {
int lin_common(bool bwd, CellArgs& a, const float* knots, int n_knots, int k, hipStream_t s) {
    KSR_CHECK_ARG(a.B > 0 && a.O > 0 && a.H > 0 && a.W > 0 && a.Kc > 0, "kan_lin: non-positive dimension");
    KSR_CHECK_ARG(a.r == 0 || a.O % (a.r * a.r) == 0, "kan_lin: out_dim %d not divisible by r^2 (r=%d)", a.O, a.r);
    a.I = 1;
    a.P = a.O * (a.Kc + 2);
    KSR_UNSUPPORTED(a.O > 64 || a.P > LIN_MAX_P, "kan_lin: out_dim %d / %d parameters per pixel exceed 64 / %d", a.O, a.P, LIN_MAX_P);
    KSR_CHECK_ARG(((reinterpret_cast<uintptr_t>(a.ly) | reinterpret_cast<uintptr_t>(a.dly)) & 15) == 0, "kan_lin: y, dy must be 16-byte aligned");
    KSR_CHECK_ARG(bwd || a.r != 4 || ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && a.out_bstride % 4 == 0),
                  "kan_lin_fwd: with shuffle_r = 4, out must be 16-byte aligned and its batch stride a multiple of 4");
    KsrKnots kn;
    int rc = fill_knots(kn, knots, n_knots, k, a.Kc);
    if (rc) return rc;
    switch (k) {
        case 0: return launch_lin<0>(bwd, a, kn, s);
        case 1: return launch_lin<1>(bwd, a, kn, s);
        case 2: return launch_lin<2>(bwd, a, kn, s);
        default: return launch_lin<3>(bwd, a, kn, s);
    }
}
}  // namespace

extern "C" int64_t ksr_kan_lin_slab_floats(int B, int H, int W, int O, int n_coef) {
    if (B <= 0 || H <= 0 || W <= 0 || O <= 0 || n_coef <= 0) return 0;
    const int rows16 = 4 * lin16_grid(B, H, W), rows = lin_grid(B, H, W);
    return (int64_t)(rows16 > rows ? rows16 : rows) * 5 * O * (n_coef + 2);
}

extern "C" int ksr_kan_lin_fwd(const float* x, const float* y, const float* lw, const float* lb, float* out, int B, int O, int H,
                               int W, int n_coef, const float* knots, int n_knots, int spline_order, int shuffle_r,
                               int64_t out_batch_stride, hipStream_t stream) {
    KSR_CHECK_ARG(x && y && lw && lb && out, "kan_lin_fwd: null pointer");
    CellArgs a{};
    a.x = x; a.ly = y; a.lw = lw; a.lb = lb; a.out = out;
    a.B = B; a.O = O; a.H = H; a.W = W; a.Kc = n_coef; a.r = shuffle_r;
    a.out_bstride = out_batch_stride > 0 ? out_batch_stride : (long)O * H * W;
    return lin_common(false, a, knots, n_knots, spline_order, stream);
}

extern "C" int ksr_kan_lin_bwd(const float* x, const float* y, const float* lw, const float* lb, const float* gout, float* dy,
                               float* slab, float* dlwb, int B, int O, int H, int W, int n_coef, const float* knots, int n_knots,
                               int spline_order, int shuffle_r, int64_t gout_batch_stride, hipStream_t stream) {
    KSR_CHECK_ARG(x && y && lw && lb && gout && dy && slab && dlwb, "kan_lin_bwd: null pointer");
    CellArgs a{};
    a.x = x; a.ly = y; a.lw = lw; a.lb = lb; a.gout = gout; a.dly = dy; a.lslab = slab;
    a.B = B; a.O = O; a.H = H; a.W = W; a.Kc = n_coef; a.r = shuffle_r;
    a.out_bstride = gout_batch_stride > 0 ? gout_batch_stride : (long)O * H * W;
    int rc = lin_common(true, a, knots, n_knots, spline_order, stream);
    if (rc) return rc;
    const int E = 5 * a.P;
    hipLaunchKernelGGL(kan_lin_reduce_kernel, dim3((unsigned)((E + 15) / 16)), dim3(256), 0, stream, slab, lin_slab_rows(lin16_fits(a), B, H, W), E, dlwb);
    KSR_LAUNCH_CHECK("kan_lin_bwd(reduce)");
    return KSR_OK;
}

extern "C" int ksr_kan_cell_bwd(const float* x, const float* w, const float* gout, float* gw, float* gx, int B, int I,
                                int O, int H, int W, int n_coef, const float* knots, int n_knots, int spline_order,
                                int w_nhwc, int shuffle_r, int64_t gout_batch_stride, hipStream_t stream) {
    KSR_CHECK_ARG(x && w && gout && gw, "kan_cell_bwd: null pointer");
    CellArgs a{};
    a.x = x; a.w = w; a.gout = gout; a.gw = gw; a.gx = gx;
    a.B = B; a.I = I; a.O = O; a.H = H; a.W = W; a.Kc = n_coef; a.w_nhwc = w_nhwc; a.r = shuffle_r;
    a.out_bstride = gout_batch_stride > 0 ? gout_batch_stride : (long)O * H * W;
    return cell_common(true, a, knots, n_knots, spline_order, stream);
}

extern "C" int ksr_bilinear_up(const float* in, float* out, int B, int n_channels, int H, int W, int r,
                               int64_t in_batch_stride, int in_channel0, int64_t out_batch_stride,
                               int out_channel0, hipStream_t stream) {
    KSR_CHECK_ARG(in && out, "bilinear_up: null pointer");
    KSR_CHECK_ARG(B > 0 && n_channels > 0 && H > 0 && W > 0 && r >= 1, "bilinear_up: bad dimension");
    BilArgs a{};
    a.in = in; a.out = out; a.B = B; a.nc = n_channels; a.H = H; a.W = W; a.r = r;
    a.in_bstride = in_batch_stride; a.in_cstride = (long)H * W;
    a.out_bstride = out_batch_stride; a.out_cstride = (long)H * r * (long)W * r;
    a.ic0 = in_channel0; a.oc0 = out_channel0;
    const bool vec = ((W * r) % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     (out_batch_stride % 4 == 0);
    const long planes = (long)B * n_channels;
    if (vec && (r == 2 || r == 4) && planes <= 65535 && (H + 3) / 4 <= 65535) {
        const dim3 grid((unsigned)((W * r / 4 + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)planes), block(64, 4);
        if (r == 4) hipLaunchKernelGGL(bilinear_up_rows_kernel<4>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(bilinear_up_rows_kernel<2>, grid, block, 0, stream, a);
    } else if (vec) {
        const long total = (long)B * n_channels * (H * r) * (long)(W * r / 4);
        hipLaunchKernelGGL(bilinear_up_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    } else {
        const long total = (long)B * n_channels * (H * r) * (long)(W * r);
        hipLaunchKernelGGL(bilinear_up_scalar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    }
    KSR_LAUNCH_CHECK("bilinear_up");
    return KSR_OK;
}

// Adjoint of ksr_bilinear_up: gin planes in_channel0.. of [B, C, H, W] (written) from gout planes out_channel0.. of the
// r x larger tensor.  Strides in elements.
extern "C" int ksr_bilinear_up_bwd(const float* gout, float* gin, int B, int n_channels, int H, int W, int r,
                                   int64_t in_batch_stride, int in_channel0, int64_t out_batch_stride, int out_channel0,
                                   hipStream_t stream) {
    KSR_CHECK_ARG(gout && gin && B > 0 && n_channels > 0 && H > 0 && W > 0 && r > 0, "bilinear_up_bwd: bad argument");
    BilArgs a{};
    a.B = B; a.nc = n_channels; a.H = H; a.W = W; a.r = r;
    a.in_bstride = in_batch_stride; a.in_cstride = (long)H * W;
    a.out_bstride = out_batch_stride; a.out_cstride = (long)H * r * W * r;
    a.ic0 = in_channel0; a.oc0 = out_channel0;
    const long total = (long)B * n_channels * H * W;
    hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, gout, gin, a);
    KSR_LAUNCH_CHECK("bilinear_up_bwd");
    return KSR_OK;
}

// gx[b, y, x] (images gx_batch_stride elements apart) += the gradient that reaches the luma plane through the first FFN's
// front, given dpre [N] (gradient of the depthwise conv's output before the ReLU, as the fn_1 backward kernels leave it).
extern "C" int ksr_stem_dx(const float* dpre, const float* pw1_w, const float* dw_w, float* gx, int64_t gx_batch_stride,
                           int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(dpre && pw1_w && dw_w && gx && B > 0 && H > 0 && W > 0, "stem_dx: bad argument");
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(stem_dx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dpre, pw1_w, dw_w, gx,
                       (long)gx_batch_stride, B, H, W);
    KSR_LAUNCH_CHECK("stem_dx");
    return KSR_OK;
}
