// Forward kernels for the blocks of the SeemoRe backbone that the reference can construct (conv_sep_kan/seemore.py:
// StripedConv2d :325-341, GatedFFN :354-382, LayerNorm :386-410, StripedConvFormer :134-155, SME :114-129, Expert :289-302,
// Router :305-318, MoELayer :261-285, MoEBlock :188-258).  Activations are NHWC fp32 ([B, H, W, C] with a row length and a
// first channel per operand, so the halves of a 2C-channel tensor are read in place: torch.chunk copies nothing here).
// These blocks are a few kflop per pixel over 64-channel tensors: bound by memory and launch latency, no matrix cores.
#include "ksr_common.h"

namespace {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// ------------------------------------------------------------------------------------------------ generic convolution
// out[b, yo, xo, co] = act(bias[co] + sum_{ky,kx,ci} W * in[b, yo*s - ph + ky, xo*s - pw + kx, ci] (* mul[...same...])) + res
//   dense:     W = w[(ky*KW + kx)*Cin*Cout + ci*Cout + co]       (host re-layout of torch's [co][ci][ky][kx])
//   depthwise: W = w[(ky*KW + kx)*Cout + co], ci = co
// A thread owns one output pixel x 8 consecutive output channels; the 8 threads of a pixel (Cout = 64) sit in neighbouring
// lanes: 256-byte output rows, inputs of a pixel shared through the cache.
struct SmConvArgs {
    const float* in; int in_ld, in_c0;
    const float* mul; int mul_ld, mul_c0;        // optional elementwise factor of the input (same pixel, same channel)
    const float* w; const float* bias;
    const float* res; int res_ld, res_c0;        // optional residual added after the activation
    float* out; int out_ld, out_c0;
    int B, H, W, Ho, Wo, Cin, Cout, KH, KW, stride, pad_h, pad_w, depthwise, act;
};

__global__ void __launch_bounds__(256) sm_conv_kernel(SmConvArgs a) {
    const int groups = a.Cout >> 3;
    const long total = (long)a.B * a.Ho * a.Wo * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % groups);
        const long pix = idx / groups;
        const int xo = (int)(pix % a.Wo), yo = (int)((pix / a.Wo) % a.Ho), b = (int)(pix / ((long)a.Wo * a.Ho));
        const int co = cg * 8;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = a.bias ? a.bias[co + j] : 0.f;
        for (int ky = 0; ky < a.KH; ++ky) {
            const int yi = yo * a.stride - a.pad_h + ky;
            if (yi < 0 || yi >= a.H) continue;
            for (int kx = 0; kx < a.KW; ++kx) {
                const int xi = xo * a.stride - a.pad_w + kx;
                if (xi < 0 || xi >= a.W) continue;
                const long ip = ((long)b * a.H + yi) * a.W + xi;
                const int tap = ky * a.KW + kx;
                if (a.depthwise) {
                    const float* xp = a.in + ip * a.in_ld + a.in_c0 + co;
                    const float* wp = a.w + (long)tap * a.Cout + co;
                    const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 4);
                    const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
                    float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    if (a.mul) {
                        const float* mp = a.mul + ip * a.mul_ld + a.mul_c0 + co;
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] *= mp[j];
                    }
                    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
                } else {
                    const float* xp = a.in + ip * a.in_ld + a.in_c0;
                    const float* mp = a.mul ? a.mul + ip * a.mul_ld + a.mul_c0 : nullptr;
                    const float* wp = a.w + (long)tap * a.Cin * a.Cout + co;
                    for (int ci = 0; ci < a.Cin; ci += 4) {
                        float4 xv = *reinterpret_cast<const float4*>(xp + ci);
                        if (mp) {
                            const float4 m = *reinterpret_cast<const float4*>(mp + ci);
                            xv.x *= m.x; xv.y *= m.y; xv.z *= m.z; xv.w *= m.w;
                        }
                        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float* wq = wp + (long)(ci + q) * a.Cout;
                            const float4 w0 = *reinterpret_cast<const float4*>(wq), w1 = *reinterpret_cast<const float4*>(wq + 4);
                            acc[0] = fmaf(xs[q], w0.x, acc[0]); acc[1] = fmaf(xs[q], w0.y, acc[1]);
                            acc[2] = fmaf(xs[q], w0.z, acc[2]); acc[3] = fmaf(xs[q], w0.w, acc[3]);
                            acc[4] = fmaf(xs[q], w1.x, acc[4]); acc[5] = fmaf(xs[q], w1.y, acc[5]);
                            acc[6] = fmaf(xs[q], w1.z, acc[6]); acc[7] = fmaf(xs[q], w1.w, acc[7]);
                        }
                    }
                }
            }
        }
        if (a.act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = gelu_erf(acc[j]);
        }
        if (a.res) {
            const float* rp = a.res + pix * a.res_ld + a.res_c0 + co;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += rp[j];
        }
        float* op = a.out + pix * a.out_ld + a.out_c0 + co;
        *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm over channels
// 8 lanes per pixel (8 channels each, C = 64); mean and biased variance through the lane group, eps inside the root.
__global__ void __launch_bounds__(256) sm_layernorm_kernel(const float* __restrict__ in, const float* __restrict__ gain,
                                                           const float* __restrict__ shift, float* __restrict__ out, long npix,
                                                           float eps) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pix = idx >> 3;
    const int c0 = (int)(idx & 7) * 8;
    const bool live = pix < npix;
    float v[8];
    {
        const float* p = in + (live ? pix : 0) * 64 + c0;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mu = s * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] -= mu; q += v[j] * v[j]; }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) q += __shfl_xor(q, off, 64);
    const float inv = 1.f / sqrtf(q * (1.f / 64.f) + eps);
    if (!live) return;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = gain[c0 + j] * (v[j] * inv) + shift[c0 + j];
    float* p = out + pix * 64 + c0;
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

// ------------------------------------------------------------------------------------------------ bilinear resize + add
// out[b, y, x, c] = res[b, y, x, c] + bilinear(src[b, :, :, c]) at (y, x), align_corners = False with the scale taken from the
// SIZES (F.interpolate(size=...)): s = in / out, src = s * (dst + 0.5) - 0.5 clamped at 0, second tap clamped at in - 1.
__global__ void __launch_bounds__(256) sm_resize_add_kernel(const float* __restrict__ src, int hs, int ws, const float* __restrict__ res,
                                                            int res_ld, int res_c0, float* __restrict__ out, int B, int H, int W) {
    const long total = (long)B * H * W * 16;          // a thread owns 4 channels of a pixel
    const float sy = (float)hs / (float)H, sx = (float)ws / (float)W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx & 15) * 4;
        const long pix = idx >> 4;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + 1 < hs ? y0 + 1 : hs - 1, x1 = x0 + 1 < ws ? x0 + 1 : ws - 1;
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* sb = src + (long)b * hs * ws * 64 + c;
        const float4 v00 = *reinterpret_cast<const float4*>(sb + ((long)y0 * ws + x0) * 64);
        const float4 v01 = *reinterpret_cast<const float4*>(sb + ((long)y0 * ws + x1) * 64);
        const float4 v10 = *reinterpret_cast<const float4*>(sb + ((long)y1 * ws + x0) * 64);
        const float4 v11 = *reinterpret_cast<const float4*>(sb + ((long)y1 * ws + x1) * 64);
        const float4 r = *reinterpret_cast<const float4*>(res + pix * res_ld + res_c0 + c);
        float4 o;
        o.x = r.x + (hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x));
        o.y = r.y + (hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y));
        o.z = r.z + (hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z));
        o.w = r.w + (hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w));
        *reinterpret_cast<float4*>(out + pix * 64 + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------ router
// One workgroup per image: channel means over the pixels (fixed order), logits = Wr mean (no bias), softmax, the topk
// largest probabilities kept (the others 0): weights [B, E].  logits_out (optional) receives the raw logits.
constexpr int SM_MAX_EXPERTS = 8;
__global__ void __launch_bounds__(256) sm_route_kernel(const float* __restrict__ in, int in_ld, int in_c0, const float* __restrict__ wr,
                                                       int E, int topk, long HW, float* __restrict__ weights,
                                                       float* __restrict__ logits_out) {
    __shared__ float part[4][64];
    __shared__ float mean[64];
    const int b = blockIdx.x, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* base = in + (long)b * HW * in_ld + in_c0 + c;
    float s0 = 0.f, s1 = 0.f;
    long p = g;
    for (; p + 4 < HW; p += 8) { s0 += base[p * in_ld]; s1 += base[(p + 4) * in_ld]; }
    for (; p < HW; p += 4) s0 += base[p * in_ld];
    part[g][c] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 64) mean[c] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / (float)HW;
    __syncthreads();
    if (threadIdx.x != 0) return;
    float logit[SM_MAX_EXPERTS], pr[SM_MAX_EXPERTS];
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) {
        float s = 0.f;
        for (int k = 0; k < 64; ++k) s = fmaf(wr[e * 64 + k], mean[k], s);
        logit[e] = s;
        mx = fmaxf(mx, s);
        if (logits_out) logits_out[b * E + e] = s;
    }
    float z = 0.f;
    for (int e = 0; e < E; ++e) { pr[e] = expf(logit[e] - mx); z += pr[e]; }
    bool taken[SM_MAX_EXPERTS];
    for (int e = 0; e < E; ++e) { pr[e] /= z; taken[e] = false; weights[b * E + e] = 0.f; }
    for (int t = 0; t < topk; ++t) {
        int best = -1;
        for (int e = 0; e < E; ++e)
            if (!taken[e] && (best < 0 || pr[e] > pr[best])) best = e;
        taken[best] = true;
        weights[b * E + best] = pr[best];
    }
}

// ------------------------------------------------------------------------------------------------ mixture of experts
// out[n] = x[n] + sum_e weights[b(n), e] * (W3_e ((W2_e k[n] + b2_e) * (W1_e x[n] + b1_e)) + b3_e): the per-pixel low-rank
// experts of an image whose router weight is not 0.  A thread owns a pixel (x, k and the sum in registers); the
// weights are uniform over the wave.
struct SmExpertTable {
    const float *w1[SM_MAX_EXPERTS], *b1[SM_MAX_EXPERTS], *w2[SM_MAX_EXPERTS], *b2[SM_MAX_EXPERTS], *w3[SM_MAX_EXPERTS], *b3[SM_MAX_EXPERTS];
    int low[SM_MAX_EXPERTS];
    int n;
};

__global__ void __launch_bounds__(64) sm_experts_kernel(const float* __restrict__ x, int x_ld, int x_c0, const float* __restrict__ k,
                                                        int k_ld, int k_c0, const float* __restrict__ weights, SmExpertTable t,
                                                        float* __restrict__ out, long HW) {
    // grid (ceil(HW / 64), B): one image per blockIdx.y, so the router weights are uniform over the workgroup
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * 64 + threadIdx.x;
    const bool live = p < HW;
    const long n = (long)b * HW + (live ? p : 0);
    float xv[64], kv[64], acc[64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(x + n * x_ld + x_c0 + 4 * q);
        const float4 c = *reinterpret_cast<const float4*>(k + n * k_ld + k_c0 + 4 * q);
        xv[4 * q] = a.x; xv[4 * q + 1] = a.y; xv[4 * q + 2] = a.z; xv[4 * q + 3] = a.w;
        kv[4 * q] = c.x; kv[4 * q + 1] = c.y; kv[4 * q + 2] = c.z; kv[4 * q + 3] = c.w;
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = xv[c];
    for (int e = 0; e < t.n; ++e) {
        const float we = weights[b * t.n + e];
        if (we == 0.f) continue;
        const float *w1 = t.w1[e], *w2 = t.w2[e], *w3 = t.w3[e];
#pragma unroll
        for (int c = 0; c < 64; ++c) acc[c] = fmaf(we, t.b3[e][c], acc[c]);
        for (int l = 0; l < t.low[e]; ++l) {
            float s1 = t.b1[e][l], s2 = t.b2[e][l];
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                s1 = fmaf(w1[l * 64 + c], xv[c], s1);
                s2 = fmaf(w2[l * 64 + c], kv[c], s2);
            }
            const float m = we * (s1 * s2);
            const int low = t.low[e];
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] = fmaf(w3[c * low + l], m, acc[c]);
        }
    }
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        *reinterpret_cast<float4*>(out + n * 64 + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

}  // namespace

// Convolution over NHWC fp32 tensors (see sm_conv_kernel).  in [B,H,W,in_ld] from channel in_c0 (Cin channels), optional mul
// (same shape rules), w in kernel layout (dense [KH*KW][Cin][Cout], depthwise [KH*KW][Cout]), bias [Cout] or NULL, act 0 = none /
// 1 = exact GELU, optional residual res [B,Ho,Wo,res_ld] from res_c0 added after the activation, out [B,Ho,Wo,out_ld] from
// out_c0, Ho = (H + 2 pad_h - KH) / stride + 1 (likewise Wo).  Cout a multiple of 8, Cin a multiple of 4, every row length and
// first channel a multiple of 4 (16-byte accesses).
extern "C" int ksr_sm_conv_fwd(const float* in, int in_ld, int in_c0, const float* mul, int mul_ld, int mul_c0, const float* w,
                               const float* bias, const float* res, int res_ld, int res_c0, float* out, int out_ld, int out_c0,
                               int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                               int depthwise, int act, hipStream_t stream) {
    KSR_CHECK_ARG(in && w && out && B > 0 && H > 0 && W > 0, "sm_conv_fwd: bad argument");
    KSR_CHECK_ARG(Cout > 0 && Cout % 8 == 0 && Cin > 0 && Cin % 4 == 0 && (!depthwise || Cin == Cout), "sm_conv_fwd: Cout %% 8, Cin %% 4, depthwise needs Cin == Cout");
    KSR_CHECK_ARG(KH > 0 && KW > 0 && stride > 0 && pad_h >= 0 && pad_w >= 0 && (act == 0 || act == 1), "sm_conv_fwd: bad kernel shape / activation");
    KSR_CHECK_ARG(in_ld % 4 == 0 && in_c0 % 4 == 0 && out_ld % 4 == 0 && out_c0 % 4 == 0 && in_c0 + Cin <= in_ld && out_c0 + Cout <= out_ld,
                  "sm_conv_fwd: row lengths / first channels must be multiples of 4 and hold the channels");
    KSR_CHECK_ARG(!mul || (mul_ld % 4 == 0 && mul_c0 % 4 == 0 && mul_c0 + Cin <= mul_ld), "sm_conv_fwd: bad mul operand");
    KSR_CHECK_ARG(!res || (res_ld % 4 == 0 && res_c0 % 4 == 0 && res_c0 + Cout <= res_ld), "sm_conv_fwd: bad residual operand");
    const int Ho = (H + 2 * pad_h - KH) / stride + 1, Wo = (W + 2 * pad_w - KW) / stride + 1;
    KSR_CHECK_ARG(Ho > 0 && Wo > 0, "sm_conv_fwd: the kernel does not fit the image");
    SmConvArgs a{in, in_ld, in_c0, mul, mul_ld, mul_c0, w, bias, res, res_ld, res_c0, out, out_ld, out_c0,
                 B, H, W, Ho, Wo, Cin, Cout, KH, KW, stride, pad_h, pad_w, depthwise, act};
    const long total = (long)B * Ho * Wo * (Cout / 8);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(sm_conv_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("sm_conv_fwd");
    return KSR_OK;
}

// LayerNorm(data_format = "channels_first") of seemore.py:386-410 on NHWC pixels: in / out [npix, 64].
extern "C" int ksr_sm_layernorm_fwd(const float* in, const float* gain, const float* shift, float* out, int64_t npix, float eps,
                                    hipStream_t stream) {
    KSR_CHECK_ARG(in && gain && shift && out && npix > 0, "sm_layernorm_fwd: bad argument");
    const long threads = (long)npix * 8;
    hipLaunchKernelGGL(sm_layernorm_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, in, gain, shift, out, (long)npix, eps);
    KSR_LAUNCH_CHECK("sm_layernorm_fwd");
    return KSR_OK;
}

// out [B,H,W,64] = res + F.interpolate(src [B,hs,ws,64], size=(H,W), mode="bilinear", align_corners=False): the calibration
// path of MoEBlock (seemore.py:236-244).
extern "C" int ksr_sm_resize_add_fwd(const float* src, int hs, int ws, const float* res, int res_ld, int res_c0, float* out, int B,
                                     int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(src && res && out && B > 0 && H > 0 && W > 0 && hs > 0 && ws > 0, "sm_resize_add_fwd: bad argument");
    KSR_CHECK_ARG(res_ld % 4 == 0 && res_c0 % 4 == 0 && res_c0 + 64 <= res_ld, "sm_resize_add_fwd: bad residual operand");
    const long total = (long)B * H * W * 16, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(sm_resize_add_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, src, hs, ws, res,
                       res_ld, res_c0, out, B, H, W);
    KSR_LAUNCH_CHECK("sm_resize_add_fwd");
    return KSR_OK;
}

// Router + top-k of MoELayer (seemore.py:305-318, :269-277): weights [B, E] = softmax(Wr mean_pixels(in)) with all but the
// topk largest entries of each image set to 0; logits_out [B, E] (optional) = the raw logits.  E <= 8, 64 channels.
extern "C" int ksr_sm_route_fwd(const float* in, int in_ld, int in_c0, const float* router_w, int E, int topk, int B, int64_t HW,
                                float* weights, float* logits_out, hipStream_t stream) {
    KSR_CHECK_ARG(in && router_w && weights && B > 0 && HW > 0, "sm_route_fwd: bad argument");
    KSR_CHECK_ARG(E > 0 && E <= SM_MAX_EXPERTS && topk > 0 && topk <= E, "sm_route_fwd: 1 <= topk <= experts <= 8");
    hipLaunchKernelGGL(sm_route_kernel, dim3((unsigned)B), dim3(256), 0, stream, in, in_ld, in_c0, router_w, E, topk, (long)HW, weights,
                       logits_out);
    KSR_LAUNCH_CHECK("sm_route_fwd");
    return KSR_OK;
}

// MoELayer's sum (seemore.py:273-285): out [B*HW, 64] = x + sum_e weights[b, e] * Expert_e(x, k) (Expert: seemore.py:289-302).
// ptrs: 6 device pointers per expert (conv_1.weight [low,64], conv_1.bias, conv_2.weight [low,64], conv_2.bias, conv_3.weight
// [64,low], conv_3.bias), host array of 6 * n_experts; low: host array of n_experts.
extern "C" int ksr_sm_experts_fwd(const float* x, int x_ld, int x_c0, const float* k, int k_ld, int k_c0, const float* weights,
                                  const float* const* ptrs, const int* low, int n_experts, float* out, int B, int64_t HW,
                                  hipStream_t stream) {
    KSR_CHECK_ARG(x && k && weights && ptrs && low && out && B > 0 && HW > 0, "sm_experts_fwd: bad argument");
    KSR_CHECK_ARG(n_experts > 0 && n_experts <= SM_MAX_EXPERTS, "sm_experts_fwd: at most 8 experts");
    KSR_CHECK_ARG(x_ld % 4 == 0 && x_c0 % 4 == 0 && k_ld % 4 == 0 && k_c0 % 4 == 0 && x_c0 + 64 <= x_ld && k_c0 + 64 <= k_ld, "sm_experts_fwd: bad operand layout");
    SmExpertTable t{};
    t.n = n_experts;
    for (int e = 0; e < n_experts; ++e) {
        for (int q = 0; q < 6; ++q) KSR_CHECK_ARG(ptrs[6 * e + q] != nullptr, "sm_experts_fwd: null parameter of expert %d", e);
        KSR_CHECK_ARG(low[e] > 0 && low[e] <= 64, "sm_experts_fwd: low dimension of expert %d out of range", e);
        t.w1[e] = ptrs[6 * e]; t.b1[e] = ptrs[6 * e + 1]; t.w2[e] = ptrs[6 * e + 2]; t.b2[e] = ptrs[6 * e + 3];
        t.w3[e] = ptrs[6 * e + 4]; t.b3[e] = ptrs[6 * e + 5];
        t.low[e] = low[e];
    }
    hipLaunchKernelGGL(sm_experts_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)B), dim3(64), 0, stream, x, x_ld, x_c0, k, k_ld, k_c0,
                       weights, t, out, (long)HW);
    KSR_LAUNCH_CHECK("sm_experts_fwd");
    return KSR_OK;
}
