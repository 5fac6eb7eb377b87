// UNet hyper-encoder, forward (conv_sep_kan/unet.py:7-42, conv_sep_kan/unet_blocks.py:9-178): the alternative to the
// GazeTR encoder that maps the luma plane to the per-pixel KAN parameters.  All activations are NHWC fp32
// ([B,H,W,C], C in {1,4,8,16,32,64,P}).  The network is ~6 kflop per pixel over tensors of 4..64 channels, i.e.
// bound by HBM traffic and launch count, not by arithmetic: the kernels are direct convolutions with one output pixel
// per lane, weights read through uniform (scalar) loads, BatchNorm statistics accumulated by the producing
// convolution and BatchNorm + ReLU folded into the loads of the consuming one.
#include "ksr_common.h"

namespace {

struct ConvArgs {
    const float* src0; const float* src1;   // [B,Hin,Win,c0], [B,Hin,Win,c1] (channel concatenation; src1 may be null)
    int c0, c1;
    const float* in_scale; const float* in_shift;   // optional: relu(x*scale[c]+shift[c]) applied to src0 on load
    const float* weight; long w_image_stride;       // [ks*ks][c0+c1][cout]; per-image weights when the stride is non-zero
    const float* bias;                              // [cout] or null
    float* out; int cout;                           // [B,Hout,Wout,cout]
    float* partials;                                // null or [gridDim.y*gridDim.x][2][cout]: sum, sum of squares
    int ks, stride, reflect;
    int Hin, Win, Hout, Wout;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int CT>
__device__ __forceinline__ void conv_taps(const float* __restrict__ src, int cs, long pix, bool ok, const float* __restrict__ sc,
                                          const float* __restrict__ sh, const float* __restrict__ wt, int cout, float (&acc)[CT]) {
    if ((cs & 3) == 0) {
        int ci = 0;
        // 16 channels at a time: the four loads are issued together, then their 16 x CT products
        for (; ci + 16 <= cs; ci += 16) {
            float4 q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const float4*>(src + pix * cs + ci + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = ci + 4 * j + e;
                    if (sc) v[e] = fmaxf(fmaf(v[e], sc[c], sh[c]), 0.f);
                    v[e] = ok ? v[e] : 0.f;
                    const float* w = wt + (long)c * cout;
#pragma unroll
                    for (int co = 0; co < CT; ++co) acc[co] = fmaf(v[e], w[co], acc[co]);
                }
            }
        }
        for (; ci < cs; ci += 4) {
            const float4 v4 = *reinterpret_cast<const float4*>(src + pix * cs + ci);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (sc) v[e] = fmaxf(fmaf(v[e], sc[ci + e], sh[ci + e]), 0.f);
                v[e] = ok ? v[e] : 0.f;
                const float* w = wt + (long)(ci + e) * cout;
#pragma unroll
                for (int co = 0; co < CT; ++co) acc[co] = fmaf(v[e], w[co], acc[co]);
            }
        }
    } else {
        for (int ci = 0; ci < cs; ++ci) {
            float v = src[pix * cs + ci];
            if (sc) v = fmaxf(fmaf(v, sc[ci], sh[ci]), 0.f);
            v = ok ? v : 0.f;
            const float* w = wt + (long)ci * cout;
#pragma unroll
            for (int co = 0; co < CT; ++co) acc[co] = fmaf(v, w[co], acc[co]);
        }
    }
}

// grid (ceil(Hout*Wout/256), B, cout/CT): one output pixel and CT output channels per lane.
// C0 > 0: the 3x3, stride-1, zero-padded conv over C0 (+ C1) input channels with everything known at compile time, for
// the shallow levels (1..16 channels at full and half resolution).  With run-time trip counts the nine taps' loads and
// the weights' uniform loads were issued and waited for one tap at a time (a 4 -> 4 conv over 8 x 256 x 256 pixels took
// 28 us for 17 MB of traffic); unrolled, they are in flight together.  Same products in the same order.
template <int CT, int C0 = 0, int C1 = 0>
__global__ void __launch_bounds__(256) unet_conv_kernel(ConvArgs a) {
    constexpr bool FIXED = C0 > 0;
    const int b = blockIdx.y, cz = blockIdx.z * CT;
    const int npix = a.Hout * a.Wout;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < npix;
    const int pc = live ? p : 0;
    const int oy = pc / a.Wout, ox = pc - oy * a.Wout;
    const int c0 = FIXED ? C0 : a.c0, c1 = FIXED ? C1 : a.c1;
    const int ks = FIXED ? 3 : a.ks, stride = FIXED ? 1 : a.stride;
    const bool reflect = FIXED ? false : a.reflect != 0;
    const int cin = c0 + c1;
    const float* wgt = a.weight + (long)b * a.w_image_stride + cz;
    float acc[CT];
#pragma unroll
    for (int co = 0; co < CT; ++co) acc[co] = a.bias ? a.bias[cz + co] : 0.f;
    const int pad = ks == 3 ? 1 : 0;
#pragma unroll
    for (int ky = 0; ky < ks; ++ky) {
        int iy = oy * stride + ky - pad;
        bool oky = true;
        if (reflect) iy = iy < 0 ? -iy : (iy >= a.Hin ? 2 * a.Hin - 2 - iy : iy);
        else oky = iy >= 0 && iy < a.Hin;
#pragma unroll
        for (int kx = 0; kx < ks; ++kx) {
            int ix = ox * stride + kx - pad;
            bool ok = oky;
            if (reflect) ix = ix < 0 ? -ix : (ix >= a.Win ? 2 * a.Win - 2 - ix : ix);
            else ok = ok && ix >= 0 && ix < a.Win;
            const long pix = ((long)b * a.Hin + (ok ? iy : 0)) * a.Win + (ok ? ix : 0);
            const float* wt = wgt + (long)(ky * ks + kx) * cin * a.cout;
            conv_taps<CT>(a.src0, c0, pix, ok, a.in_scale, a.in_shift, wt, a.cout, acc);
            if (FIXED ? C1 > 0 : a.src1 != nullptr) conv_taps<CT>(a.src1, c1, pix, ok, nullptr, nullptr, wt + (long)c0 * a.cout, a.cout, acc);
        }
    }
    if (live) {
        float* o = a.out + ((long)b * npix + p) * a.cout + cz;
#pragma unroll
        for (int co = 0; co < CT; co += 4) *reinterpret_cast<float4*>(o + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
    }
    if (a.partials) {
        __shared__ float red[4][2 * CT];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int co = 0; co < CT; ++co) {
            const float v = live ? acc[co] : 0.f;
            const float s = wave_sum(v), q = wave_sum(v * v);
            if (lane == 0) { red[wave][co] = s; red[wave][CT + co] = q; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * CT) {
            const int t = threadIdx.x;
            const float v = blockDim.x == 64 ? red[0][t] : (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
            const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
            a.partials[(wg * 2 + t / CT) * a.cout + cz + t % CT] = v;
        }
    }
}

// Small images with many channels (the deep levels: 32x32 ... 1x1 pixels, 16..128 input channels): one pixel per lane
// would leave a few thousand lanes with serial dot products of up to 1152 terms whose weights (each used once per wave)
// miss the scalar cache one by one.  Here LPP lanes share a pixel and split its input channels, the weights of one
// kernel row are staged in LDS ([kx][e][co4][g][4] for channel 4g+e: the lanes of a pixel read consecutive 16-byte
// words), and the partial sums meet in a butterfly.  16 output channels per lane group; grid (ceil(npix/PPW), B, cout/16).
template <int LPP>
__global__ void __launch_bounds__(256) unet_conv_ksplit_kernel(ConvArgs a) {
    constexpr int CT = 16, PPW = 256 / LPP;
    __shared__ __attribute__((aligned(16))) float wl[3 * 128 * CT];
    __shared__ float sred[PPW][CT];
    const int b = blockIdx.y, cz = blockIdx.z * CT;
    const int npix = a.Hout * a.Wout;
    const int l = threadIdx.x % LPP, pslot = threadIdx.x / LPP;
    const int p = blockIdx.x * PPW + pslot;
    const bool live = p < npix;
    const int pc = live ? p : 0;
    const int oy = pc / a.Wout, ox = pc - oy * a.Wout;
    const int cin = a.c0 + a.c1, G = cin >> 2;
    const int pad = a.ks == 3 ? 1 : 0;
    float acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = 0.f;
    for (int ky = 0; ky < a.ks; ++ky) {
        __syncthreads();
        for (int i = threadIdx.x; i < a.ks * cin * 4; i += 256) {
            const int co4 = i & 3, r = i >> 2, kx = r / cin, c = r - kx * cin;
            const float4 w4 = *reinterpret_cast<const float4*>(a.weight + ((long)(ky * a.ks) * cin + r) * a.cout + cz + co4 * 4);
            *reinterpret_cast<float4*>(wl + ((((kx * 4 + (c & 3)) * 4 + co4) * G + (c >> 2)) << 2)) = w4;
        }
        __syncthreads();
        int iy = oy * a.stride + ky - pad;
        bool oky = true;
        if (a.reflect) iy = iy < 0 ? -iy : (iy >= a.Hin ? 2 * a.Hin - 2 - iy : iy);
        else oky = iy >= 0 && iy < a.Hin;
        for (int kx = 0; kx < a.ks; ++kx) {
            int ix = ox * a.stride + kx - pad;
            bool ok = oky;
            if (a.reflect) ix = ix < 0 ? -ix : (ix >= a.Win ? 2 * a.Win - 2 - ix : ix);
            else ok = ok && ix >= 0 && ix < a.Win;
            const long pix = ((long)b * a.Hin + (ok ? iy : 0)) * a.Win + (ok ? ix : 0);
            for (int c = 4 * l; c < cin; c += 4 * LPP) {
                const bool second = c >= a.c0;
                const float* src = second ? a.src1 + pix * a.c1 + (c - a.c0) : a.src0 + pix * a.c0 + c;
                const float4 v4 = *reinterpret_cast<const float4*>(src);
                float v[4] = {v4.x, v4.y, v4.z, v4.w};
                if (a.in_scale && !second) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + c), t4 = *reinterpret_cast<const float4*>(a.in_shift + c);
                    v[0] = fmaxf(fmaf(v[0], s4.x, t4.x), 0.f); v[1] = fmaxf(fmaf(v[1], s4.y, t4.y), 0.f);
                    v[2] = fmaxf(fmaf(v[2], s4.z, t4.z), 0.f); v[3] = fmaxf(fmaf(v[3], s4.w, t4.w), 0.f);
                }
                const int g = c >> 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ve = ok ? v[e] : 0.f;
#pragma unroll
                    for (int co4 = 0; co4 < 4; ++co4) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wl + ((((kx * 4 + e) * 4 + co4) * G + g) << 2));
                        acc[4 * co4 + 0] = fmaf(ve, w4.x, acc[4 * co4 + 0]); acc[4 * co4 + 1] = fmaf(ve, w4.y, acc[4 * co4 + 1]);
                        acc[4 * co4 + 2] = fmaf(ve, w4.z, acc[4 * co4 + 2]); acc[4 * co4 + 3] = fmaf(ve, w4.w, acc[4 * co4 + 3]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
#pragma unroll
        for (int m = LPP >> 1; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m, 64);
        if (a.bias) acc[i] += a.bias[cz + i];
    }
    if (live && l == 0) {
        float* o = a.out + ((long)b * npix + p) * a.cout + cz;
#pragma unroll
        for (int i = 0; i < CT; i += 4) *reinterpret_cast<float4*>(o + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
    }
    if (a.partials) {
        if (l == 0) {
#pragma unroll
            for (int i = 0; i < CT; ++i) sred[pslot][i] = live ? acc[i] : 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 2 * CT) {
            const int k = threadIdx.x / CT, c = threadIdx.x % CT;
            float s = 0.f;
            for (int q = 0; q < PPW; ++q) { const float v = sred[q][c]; s += k ? v * v : v; }
            const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
            a.partials[(wg * 2 + k) * a.cout + cz + c] = s;
        }
    }
}

struct BnArgs {
    const float* partials; int nparts; long count;
    const float* gamma; const float* beta; float* rmean; float* rvar;
    float momentum, eps; int train;
    float* scale; float* shift; float* saved; int C;
};

// one workgroup per channel: batch statistics (train) or running statistics (eval) -> y = x*scale + shift
__global__ void __launch_bounds__(256) unet_bn_finalize_kernel(BnArgs a) {
    const int c = blockIdx.x;
    __shared__ double red[2][256];
    double mean, var;
    if (a.train) {
        double s = 0.0, q = 0.0;
        for (int i = threadIdx.x; i < a.nparts; i += 256) {
            s += (double)a.partials[((long)i * 2) * a.C + c];
            q += (double)a.partials[((long)i * 2 + 1) * a.C + c];
        }
        red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if ((int)threadIdx.x < m) { red[0][threadIdx.x] += red[0][threadIdx.x + m]; red[1][threadIdx.x] += red[1][threadIdx.x + m]; }
            __syncthreads();
        }
        mean = red[0][0] / (double)a.count;
        var = red[1][0] / (double)a.count - mean * mean;
        var = var > 0.0 ? var : 0.0;
    } else {
        mean = (double)a.rmean[c]; var = (double)a.rvar[c];
    }
    if (threadIdx.x == 0) {
        if (a.train) {
            const double unbiased = a.count > 1 ? var * (double)a.count / (double)(a.count - 1) : var;
            a.rmean[c] = (float)((1.0 - a.momentum) * (double)a.rmean[c] + a.momentum * mean);
            a.rvar[c] = (float)((1.0 - a.momentum) * (double)a.rvar[c] + a.momentum * unbiased);
        }
        const float sc = a.gamma[c] * (float)(1.0 / sqrt(var + (double)a.eps));
        a.scale[c] = sc;
        a.shift[c] = a.beta[c] - (float)mean * sc;
        if (a.saved) { a.saved[c] = (float)mean; a.saved[a.C + c] = (float)(1.0 / sqrt(var + (double)a.eps)); }
    }
}

struct ActArgs { const float* in; const float* scale; const float* shift; float* out; float* pooled; int B, H, W, C; };

__device__ __forceinline__ float4 bn_relu4(float4 v, float4 s, float4 t) {
    return make_float4(fmaxf(fmaf(v.x, s.x, t.x), 0.f), fmaxf(fmaf(v.y, s.y, t.y), 0.f), fmaxf(fmaf(v.z, s.z, t.z), 0.f),
                       fmaxf(fmaf(v.w, s.w, t.w), 0.f));
}

__global__ void __launch_bounds__(256) unet_bn_relu_kernel(ActArgs a) {
    const int c4n = a.C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.B * a.H * a.W * c4n) return;
    const int c = (int)(idx % c4n) * 4;
    const float4 s = *reinterpret_cast<const float4*>(a.scale + c), t = *reinterpret_cast<const float4*>(a.shift + c);
    reinterpret_cast<float4*>(a.out)[idx] = bn_relu4(reinterpret_cast<const float4*>(a.in)[idx], s, t);
}

// one 2x2 block of pixels and 4 channels per lane: activated tensor at full size plus its 2x2 maximum.  Odd sizes: the
// last row / column belongs to no pooling window (MaxPool2d floors) and only gets its activation written.
__global__ void __launch_bounds__(256) unet_bn_relu_pool_kernel(ActArgs a) {
    const int c4n = a.C >> 2, Hb = (a.H + 1) >> 1, Wb = (a.W + 1) >> 1, Hp = a.H >> 1, Wp = a.W >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.B * Hb * Wb * c4n) return;
    const int c = (int)(idx % c4n) * 4;
    long r = idx / c4n;
    const int px = (int)(r % Wb); r /= Wb;
    const int py = (int)(r % Hb);
    const int b = (int)(r / Hb);
    const float4 s = *reinterpret_cast<const float4*>(a.scale + c), t = *reinterpret_cast<const float4*>(a.shift + c);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);      // activations are >= 0
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            if (2 * py + dy >= a.H || 2 * px + dx >= a.W) continue;
            const long o = (((long)b * a.H + 2 * py + dy) * a.W + 2 * px + dx) * a.C + c;
            const float4 v = bn_relu4(*reinterpret_cast<const float4*>(a.in + o), s, t);
            *reinterpret_cast<float4*>(a.out + o) = v;
            m = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
        }
    if (py < Hp && px < Wp) *reinterpret_cast<float4*>(a.pooled + (((long)b * Hp + py) * Wp + px) * a.C + c) = m;
}

struct ConvTArgs { const float* in; const float* weight; const float* bias; float* out; int cin, cout, h, w; };

// ConvTranspose2d(kernel 2, stride 2): one input pixel -> its 2x2 output block, CT output channels per lane.
// weight [dy*2+dx][cin][cout]
template <int CT>
__global__ void __launch_bounds__(256) unet_convt_kernel(ConvTArgs a) {
    const int b = blockIdx.y, cz = blockIdx.z * CT;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.h * a.w) return;
    const int y = p / a.w, x = p - y * a.w;
    float acc[4][CT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int co = 0; co < CT; ++co) acc[q][co] = a.bias[cz + co];
    const float* src = a.in + ((long)b * a.h * a.w + p) * a.cin;
    for (int ci = 0; ci < a.cin; ci += 4) {
        const float4 v4 = *reinterpret_cast<const float4*>(src + ci);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* w = a.weight + ((long)q * a.cin + ci + e) * a.cout + cz;
#pragma unroll
                for (int co = 0; co < CT; ++co) acc[q][co] = fmaf(v[e], w[co], acc[q][co]);
            }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* o = a.out + ((((long)b * 2 * a.h) + 2 * y + (q >> 1)) * (2 * a.w) + 2 * x + (q & 1)) * a.cout + cz;
#pragma unroll
        for (int co = 0; co < CT; co += 4) *reinterpret_cast<float4*>(o + co) = make_float4(acc[q][co], acc[q][co + 1], acc[q][co + 2], acc[q][co + 3]);
    }
}

// ---- anchor attention (unet_blocks.py:131-178).  z [B,n,zc] holds q (c channels) | k (c) | a (c/2) at half resolution;
// its per-image Gram matrix z^T z [zc][zc] (the weight-gradient kernel computes it) holds q_i.a_j, a_j.k_i and the
// squared norms.
// entry e of the packed products -> the pair of channels of z: q_i a_j, then a_j k_i, then the squared norms
__device__ __forceinline__ void gram_entry(int e, int c, int& A, int& Bc) {
    const int ca = c >> 1, nqa = c * ca;
    if (e < nqa) { A = e / ca; Bc = 2 * c + e % ca; }
    else if (e < 2 * nqa) { const int f = e - nqa; A = 2 * c + f / c; Bc = c + f % c; }
    else { A = Bc = e - 2 * nqa; }
}

struct MixArgs {
    const float* gram; int zc, c, heads;       // [B][zc][zc]
    const float* temp_a; const float* temp_v;   // [heads]
    const float* wv; const float* wp;           // v_proj, project_out weights [c][c] (out, in)
    float vscale;
    float* T;                                   // [B][c (in)][c (out)]: per-image 1x1 weights, x -> attention output
};

// one workgroup per image: softmaxes of the two anchor products, their product per head, folded with v_proj and
// project_out into one c x c matrix per image
__global__ void __launch_bounds__(256) unet_att_mix_kernel(MixArgs a) {
    const int b = blockIdx.x, c = a.c, ca = c >> 1, nqa = c * ca;
    const int E = c * c + (5 * c) / 2;
    const int ch = c / a.heads, cah = ca / a.heads;
    __shared__ float g[32 * 32 + 80], at_a[32 * 16], at_k[16 * 32], M[32 * 32], T1[32 * 32];
    for (int e = threadIdx.x; e < E; e += 256) {
        int A, Bc;
        gram_entry(e, c, A, Bc);
        g[e] = a.gram[((long)b * a.zc + A) * a.zc + Bc];
    }
    __syncthreads();
    const float* nrm2 = g + 2 * nqa;              // squared norms: q [0,c), k [c,2c), a [2c,2c+ca)
    const int t = threadIdx.x;
    if (t < c) {                                   // attn_a row i = t: softmax over the anchors of its head
        const int i = t, h = i / ch;
        const float nq = fmaxf(sqrtf(nrm2[i]), 1e-12f);
        float mx = -INFINITY;
        for (int jj = 0; jj < cah; ++jj) {
            const int j = h * cah + jj;
            const float v = g[i * ca + j] / (nq * fmaxf(sqrtf(nrm2[2 * c + j]), 1e-12f)) * a.temp_a[h];
            at_a[i * ca + j] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
        for (int jj = 0; jj < cah; ++jj) { const int j = h * cah + jj; const float ex = expf(at_a[i * ca + j] - mx); at_a[i * ca + j] = ex; sum += ex; }
        for (int jj = 0; jj < cah; ++jj) at_a[i * ca + h * cah + jj] /= sum;
    } else if (t >= 64 && t < 64 + ca) {           // attn_k row j: softmax over the k channels of its head
        const int j = t - 64, h = j / cah;
        const float na = fmaxf(sqrtf(nrm2[2 * c + j]), 1e-12f);
        float mx = -INFINITY;
        for (int ii = 0; ii < ch; ++ii) {
            const int i = h * ch + ii;
            const float v = g[nqa + j * c + i] / (na * fmaxf(sqrtf(nrm2[c + i]), 1e-12f)) * a.temp_v[h];
            at_k[j * c + i] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
        for (int ii = 0; ii < ch; ++ii) { const int i = h * ch + ii; const float ex = expf(at_k[j * c + i] - mx); at_k[j * c + i] = ex; sum += ex; }
        for (int ii = 0; ii < ch; ++ii) at_k[j * c + h * ch + ii] /= sum;
    }
    __syncthreads();
    for (int idx = t; idx < c * c; idx += 256) {   // M = attn_a attn_k, block diagonal over heads
        const int i = idx / c, i2 = idx % c, h = i / ch;
        float s = 0.f;
        if (i2 / ch == h)
            for (int jj = 0; jj < cah; ++jj) s = fmaf(at_a[i * ca + h * cah + jj], at_k[(h * cah + jj) * c + i2], s);
        M[idx] = s;
    }
    __syncthreads();
    for (int idx = t; idx < c * c; idx += 256) {   // T1 = M (vscale Wv)
        const int i = idx / c, ci = idx % c, h = i / ch;
        float s = 0.f;
        for (int ii = 0; ii < ch; ++ii) s = fmaf(M[i * c + h * ch + ii], a.wv[(h * ch + ii) * c + ci], s);
        T1[idx] = s * a.vscale;
    }
    __syncthreads();
    for (int idx = t; idx < c * c; idx += 256) {   // T = Wp T1, stored [in][out]
        const int ci = idx / c, co = idx % c;
        float s = 0.f;
        for (int i = 0; i < c; ++i) s = fmaf(a.wp[co * c + i], T1[i * c + ci], s);
        a.T[(long)b * c * c + idx] = s;
    }
}

template <int CT>
int launch_conv(const ConvArgs& a, int B, int block, hipStream_t stream) {
    const int npix = a.Hout * a.Wout;
    const dim3 grid((unsigned)((npix + block - 1) / block), (unsigned)B, (unsigned)(a.cout / CT));
    const bool fixed = a.ks == 3 && a.stride == 1 && !a.reflect && a.w_image_stride == 0;
#define KSR_CONV_FIXED(c0_, c1_)                                                                              \
    if (fixed && a.c0 == c0_ && a.c1 == c1_) {                                                                \
        hipLaunchKernelGGL((unet_conv_kernel<CT, c0_, c1_>), grid, dim3(block), 0, stream, a);                \
        KSR_LAUNCH_CHECK("unet_conv");                                                                        \
        return KSR_OK;                                                                                        \
    }
    if constexpr (CT <= 8) {
        KSR_CONV_FIXED(1, 0) KSR_CONV_FIXED(4, 0) KSR_CONV_FIXED(8, 0) KSR_CONV_FIXED(4, 4)
    }
#undef KSR_CONV_FIXED
    hipLaunchKernelGGL(unet_conv_kernel<CT>, grid, dim3(block), 0, stream, a);
    KSR_LAUNCH_CHECK("unet_conv");
    return KSR_OK;
}

}  // namespace

// Lanes per workgroup: small images get one-wave workgroups so that the deep, narrow levels still fill the chip.
static int conv_block(int B, int npix) { return (long)B * npix <= 16384 ? 64 : 256; }

// Lanes per pixel of the channel-split kernel, or 0 when the pixel-per-lane kernel is the one to use.
static int conv_ksplit_lanes(int B, int npix, int c0, int c1, int cout, int64_t w_image_stride) {
    const int cin = c0 + c1;
    if (w_image_stride != 0 || cout % 16 != 0 || (long)B * npix > 32768 || (c0 & 3) || (c1 & 3)) return 0;
    if (cin != 16 && cin != 32 && cin != 64 && cin != 128) return 0;
    return cin == 16 ? 4 : (cin == 32 ? 8 : 16);
}

extern "C" int64_t ksr_unet_conv_partials(int B, int Hout, int Wout, int cin, int cout) {
    const int npix = Hout * Wout;
    const int lpp = conv_ksplit_lanes(B, npix, cin, 0, cout, 0);
    const int per_wg = lpp ? 256 / lpp : conv_block(B, npix);
    return (int64_t)B * ((npix + per_wg - 1) / per_wg);
}

extern "C" int ksr_unet_conv_fwd(const float* src0, int c0, const float* src1, int c1, const float* in_scale, const float* in_shift,
                                 const float* weight, int64_t w_image_stride, const float* bias, float* out, int cout,
                                 float* stat_partials, int ks, int stride, int reflect, int B, int Hin, int Win, hipStream_t stream) {
    KSR_CHECK_ARG(src0 && weight && out && B > 0 && Hin > 0 && Win > 0 && c0 > 0 && c1 >= 0 && cout > 0, "unet_conv: bad argument");
    KSR_CHECK_ARG((src1 != nullptr) == (c1 > 0), "unet_conv: src1 and c1 disagree");
    KSR_CHECK_ARG((in_scale != nullptr) == (in_shift != nullptr), "unet_conv: in_scale and in_shift go together");
    KSR_UNSUPPORTED(ks < 1 || ks > 3, "unet_conv: kernel size %d (1, 2 or 3)", ks);
    KSR_UNSUPPORTED(stride != 1 && stride != 2, "unet_conv: stride %d (1 or 2)", stride);
    KSR_UNSUPPORTED(cout % 4 != 0, "unet_conv: cout = %d must be a multiple of 4", cout);
    KSR_UNSUPPORTED(c1 > 0 && ((c0 & 3) || (c1 & 3)), "unet_conv: concatenated inputs need channel counts that are multiples of 4");
    KSR_UNSUPPORTED(reflect && (Hin < 2 || Win < 2), "unet_conv: reflect padding needs at least 2 pixels per side");
    KSR_UNSUPPORTED(B > 65535 || cout / 4 > 65535, "unet_conv: batch %d too large for one launch", B);
    const int pad = ks == 3 ? 1 : 0;
    ConvArgs a{src0, src1, c0, c1, in_scale, in_shift, weight, (long)w_image_stride, bias, out, cout, stat_partials, ks, stride, reflect,
               Hin, Win, (Hin + 2 * pad - ks) / stride + 1, (Win + 2 * pad - ks) / stride + 1};
    if (const int lpp = conv_ksplit_lanes(B, a.Hout * a.Wout, c0, c1, cout, w_image_stride)) {
        const int ppw = 256 / lpp;
        const dim3 grid((unsigned)((a.Hout * a.Wout + ppw - 1) / ppw), (unsigned)B, (unsigned)(cout / 16));
        if (lpp == 4) hipLaunchKernelGGL(unet_conv_ksplit_kernel<4>, grid, dim3(256), 0, stream, a);
        else if (lpp == 8) hipLaunchKernelGGL(unet_conv_ksplit_kernel<8>, grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(unet_conv_ksplit_kernel<16>, grid, dim3(256), 0, stream, a);
        KSR_LAUNCH_CHECK("unet_conv_ksplit");
        return KSR_OK;
    }
    // output channels per lane: as many as keep at least 512 workgroups in flight (the weights are uniform loads, the
    // input is re-read from L1/L2 once per channel group)
    const int npix = a.Hout * a.Wout, block = conv_block(B, npix);
    const long wgs = (long)B * ((npix + block - 1) / block);
    if (cout % 16 == 0 && wgs * (cout / 16) >= 512) return launch_conv<16>(a, B, block, stream);
    if (cout % 8 == 0 && wgs * (cout / 8) >= 512) return launch_conv<8>(a, B, block, stream);
    return launch_conv<4>(a, B, block, stream);
}

extern "C" int ksr_unet_bn_finalize(const float* stat_partials, int n_partials, int64_t count, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, int train, float* scale,
                                    float* shift, float* saved_mean_invstd, int C, hipStream_t stream) {
    KSR_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "unet_bn_finalize: bad argument");
    KSR_CHECK_ARG(!train || (stat_partials && n_partials > 0 && count > 0), "unet_bn_finalize: training mode needs the partial sums");
    BnArgs a{stat_partials, n_partials, (long)count, gamma, beta, running_mean, running_var, momentum, eps, train, scale, shift, saved_mean_invstd, C};
    hipLaunchKernelGGL(unet_bn_finalize_kernel, dim3((unsigned)C), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("unet_bn_finalize");
    return KSR_OK;
}

extern "C" int ksr_unet_bn_relu_fwd(const float* raw, const float* scale, const float* shift, float* out, float* pooled, int B, int H,
                                    int W, int C, hipStream_t stream) {
    KSR_CHECK_ARG(raw && scale && shift && out && B > 0 && H > 0 && W > 0 && C > 0, "unet_bn_relu: bad argument");
    KSR_UNSUPPORTED(C % 4 != 0, "unet_bn_relu: C = %d must be a multiple of 4", C);
    KSR_UNSUPPORTED(pooled && (H < 2 || W < 2), "unet_bn_relu: 2x2 pooling needs at least 2x2 pixels");
    ActArgs a{raw, scale, shift, out, pooled, B, H, W, C};
    if (pooled) {
        const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
        hipLaunchKernelGGL(unet_bn_relu_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    } else {
        const long total = (long)B * H * W * (C / 4);
        hipLaunchKernelGGL(unet_bn_relu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    }
    KSR_LAUNCH_CHECK("unet_bn_relu");
    return KSR_OK;
}

extern "C" int ksr_unet_convt_fwd(const float* in, const float* weight, const float* bias, float* out, int cin, int cout, int B, int h,
                                  int w, hipStream_t stream) {
    KSR_CHECK_ARG(in && weight && bias && out && B > 0 && h > 0 && w > 0, "unet_convt: bad argument");
    KSR_UNSUPPORTED(cin % 4 != 0 || cout % 4 != 0, "unet_convt: channel counts (%d, %d) must be multiples of 4", cin, cout);
    KSR_UNSUPPORTED(B > 65535, "unet_convt: batch %d too large for one launch", B);
    ConvTArgs a{in, weight, bias, out, cin, cout, h, w};
    const unsigned gx = (unsigned)((h * w + 255) / 256);
    if (cout % 8 == 0) hipLaunchKernelGGL(unet_convt_kernel<8>, dim3(gx, (unsigned)B, (unsigned)(cout / 8)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(unet_convt_kernel<4>, dim3(gx, (unsigned)B, (unsigned)(cout / 4)), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("unet_convt");
    return KSR_OK;
}

extern "C" int ksr_unet_att_mix(const float* gram, int zc, int c, int heads, int B, const float* temp_a, const float* temp_v,
                                const float* wv, const float* wp, float vscale, float* T, hipStream_t stream) {
    KSR_CHECK_ARG(gram && temp_a && temp_v && wv && wp && T && B > 0, "unet_att_mix: bad argument");
    KSR_UNSUPPORTED(c < 4 || c > 32 || (c & (c - 1)), "unet_att_mix: dim = %d (4, 8, 16 or 32)", c);
    KSR_UNSUPPORTED(heads < 1 || (c / 2) % heads != 0, "unet_att_mix: %d heads do not divide dim/2 = %d", heads, c / 2);
    KSR_CHECK_ARG(zc >= 2 * c + c / 2, "unet_att_mix: z needs at least 2.5*dim channels");
    MixArgs m{gram, zc, c, heads, temp_a, temp_v, wv, wp, vscale, T};
    hipLaunchKernelGGL(unet_att_mix_kernel, dim3((unsigned)B), dim3(256), 0, stream, m);
    KSR_LAUNCH_CHECK("unet_att_mix");
    return KSR_OK;
}
