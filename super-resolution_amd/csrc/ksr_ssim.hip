// SSIM term of the reference's training loss (src/ml/pipelines/default.py:79-81: loss = L1(Y) + 0.15 (1 - SSIM)),
// SSIM = torchmetrics' StructuralSimilarityIndexMeasure(data_range=(0, 1)) with its defaults, restated from the
// published definition (torchmetrics is not part of the reference tree):
//   inputs clamped to [0, 1]; Gaussian window 11 x 11, sigma 1.5 (separable, normalised); c1 = 0.01^2, c2 = 0.03^2;
//   mu_p, mu_t, E[p^2], E[t^2], E[pt] by windowed averaging; variances clamped at 0;
//   S = (2 mu_p mu_t + c1)(2 cov + c2) / ((mu_p^2 + mu_t^2 + c1)(var_p + var_t + c2));
//   only windows that lie entirely inside the image count (the (H-10) x (W-10) interior); mean over pixels,
//   channels and images.
// Forward: one 32 x 32 tile of windows per workgroup, separable filtering through LDS, per-workgroup partial sums
// reduced in a fixed order.  Backward (luma plane only: the chroma planes do not depend on the parameters): the
// forward also writes dS/dmu_p, dS/dE[p^2], dS/dE[pt] per window; the gradient kernel filters those three maps
// with the same window and combines them with p and t.
#include "ksr_common.h"

namespace {

constexpr int SS_K = 11, SS_R = 5, SS_T = 32, SS_IN = SS_T + SS_K - 1;   // 42

struct SsimWin { float g[SS_K]; };

struct SsimArgs {
    const float* pred; long pred_bs;     // [B, C, H, W] with images pred_bs elements apart
    const float* tgt; long tgt_bs;
    float* maps;                         // [3][B][Hi][Wi] or null: derivative maps of channel 0
    float* slab;
    int C, B, H, W;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__global__ void __launch_bounds__(256) ssim_fwd_kernel(SsimArgs a, SsimWin w) {
    __shared__ float ps[SS_IN][SS_IN + 1], ts[SS_IN][SS_IN + 1];
    __shared__ float hs[5][SS_IN][SS_T + 1];
    __shared__ float red[4];
    const int Hi = a.H - 2 * SS_R, Wi = a.W - 2 * SS_R;
    const int tiles_x = (Wi + SS_T - 1) / SS_T;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int c = blockIdx.y, b = blockIdx.z;
    const int y0 = ty * SS_T, x0 = tx * SS_T;                  // first window of the tile = first pixel it reads
    const long plane = (long)a.H * a.W;
    const float* P = a.pred + (long)b * a.pred_bs + (long)c * plane;
    const float* T = a.tgt + (long)b * a.tgt_bs + (long)c * plane;
    for (int i = threadIdx.x; i < SS_IN * SS_IN; i += 256) {
        const int r = i / SS_IN, q = i - r * SS_IN;
        const int y = y0 + r, x = x0 + q;
        const bool in = y < a.H && x < a.W;
        ps[r][q] = in ? clamp01(P[(long)y * a.W + x]) : 0.f;
        ts[r][q] = in ? clamp01(T[(long)y * a.W + x]) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_IN * SS_T; i += 256) {
        const int r = i / SS_T, q = i - r * SS_T;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float p = ps[r][q + k], t = ts[r][q + k], g = w.g[k];
            s0 += g * p; s1 += g * t; s2 += g * p * p; s3 += g * t * t; s4 += g * p * t;
        }
        hs[0][r][q] = s0; hs[1][r][q] = s1; hs[2][r][q] = s2; hs[3][r][q] = s3; hs[4][r][q] = s4;
    }
    __syncthreads();
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float sum = 0.f;
    for (int i = threadIdx.x; i < SS_T * SS_T; i += 256) {
        const int r = i / SS_T, q = i - r * SS_T;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float g = w.g[k];
            m1 += g * hs[0][r + k][q]; m2 += g * hs[1][r + k][q]; e11 += g * hs[2][r + k][q];
            e22 += g * hs[3][r + k][q]; e12 += g * hs[4][r + k][q];
        }
        const int uy = y0 + r, ux = x0 + q;
        if (uy >= Hi || ux >= Wi) continue;
        const float vp_raw = e11 - m1 * m1, vt_raw = e22 - m2 * m2;
        const float vp = fmaxf(vp_raw, 0.f), vt = fmaxf(vt_raw, 0.f);
        const float cov = e12 - m1 * m2;
        const float N1 = 2.f * m1 * m2 + c1, D1 = m1 * m1 + m2 * m2 + c1;
        const float N2 = 2.f * cov + c2, D2 = vp + vt + c2;
        const float inv = 1.f / (D1 * D2);
        const float S = N1 * N2 * inv;
        sum += S;
        if (a.maps && c == 0) {
            const float vp_on = vp_raw > 0.f ? 1.f : 0.f;              // derivative of the clamp at 0
            const float dC = 2.f * N1 * inv;                            // dS/dE[pt]
            const float dB = -S / D2 * vp_on;                           // dS/dE[p^2]
            const float dA = (2.f * m2 * N2 - 2.f * m2 * N1) * inv - S * (2.f * m1) / D1 + S * (2.f * m1) / D2 * vp_on;
            const long mi = ((long)b * Hi + uy) * Wi + ux;
            const long ms = (long)a.B * Hi * Wi;
            a.maps[mi] = dA; a.maps[ms + mi] = dB; a.maps[2 * ms + mi] = dC;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
        a.slab[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// One workgroup: strided partial sums in double, then a fixed-order tree (deterministic).
__global__ void __launch_bounds__(1024) ssim_finish_kernel(const float* slab, long G, double scale, float* out) {
    __shared__ double part[1024];
    double s = 0.0;
    for (long i = threadIdx.x; i < G; i += 1024) s += (double)slab[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(part[0] * scale);
}

struct SsimGradArgs {
    const float* maps;                   // [3][B][Hi][Wi]
    const float* pred; long pred_bs;     // luma planes
    const float* tgt; long tgt_bs;
    float* grad; long grad_bs;           // accumulated into
    int B, H, W;
    float scale;
};

// grad[x] += scale * clampmask(p[x]) * sum over windows u covering x of g(x - u) (A_u + 2 p_x B_u + t_x C_u)
__global__ void __launch_bounds__(256) ssim_grad_kernel(SsimGradArgs a, SsimWin w) {
    __shared__ float ms[3][SS_IN][SS_IN + 1];
    __shared__ float hs[3][SS_IN][SS_T + 1];
    const int Hi = a.H - 2 * SS_R, Wi = a.W - 2 * SS_R;
    const int tiles_x = (a.W + SS_T - 1) / SS_T;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.z;
    const int py0 = ty * SS_T, px0 = tx * SS_T;
    const long msz = (long)a.B * Hi * Wi;
    // local index j <-> window u = p0 - 10 + j; pixel x = p0 + q is covered by windows j = q .. q + 10
    for (int i = threadIdx.x; i < SS_IN * SS_IN; i += 256) {
        const int r = i / SS_IN, q = i - r * SS_IN;
        const int uy = py0 - (SS_K - 1) + r, ux = px0 - (SS_K - 1) + q;
        const bool in = uy >= 0 && uy < Hi && ux >= 0 && ux < Wi;
        const long mi = ((long)b * Hi + uy) * Wi + ux;
#pragma unroll
        for (int m = 0; m < 3; ++m) ms[m][r][q] = in ? a.maps[m * msz + mi] : 0.f;
    }
    __syncthreads();
    // pixel x sits at tap k = x - u of window u: weight g[k] = g[10 - k] (symmetric), so the sums below are plain
    // sliding sums over the 11 windows j = q .. q + 10
    for (int i = threadIdx.x; i < SS_IN * SS_T; i += 256) {
        const int r = i / SS_T, q = i - r * SS_T;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float g = w.g[SS_K - 1 - k];
            s0 += g * ms[0][r][q + k]; s1 += g * ms[1][r][q + k]; s2 += g * ms[2][r][q + k];
        }
        hs[0][r][q] = s0; hs[1][r][q] = s1; hs[2][r][q] = s2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_T * SS_T; i += 256) {
        const int r = i / SS_T, q = i - r * SS_T;
        const int y = py0 + r, x = px0 + q;
        if (y >= a.H || x >= a.W) continue;
        float gA = 0.f, gB = 0.f, gC = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float g = w.g[SS_K - 1 - k];
            gA += g * hs[0][r + k][q]; gB += g * hs[1][r + k][q]; gC += g * hs[2][r + k][q];
        }
        const long off = (long)y * a.W + x;
        const float p = a.pred[(long)b * a.pred_bs + off], t = clamp01(a.tgt[(long)b * a.tgt_bs + off]);
        const float inside = (p >= 0.f && p <= 1.f) ? 1.f : 0.f;
        a.grad[(long)b * a.grad_bs + off] += a.scale * inside * (gA + 2.f * clamp01(p) * gB + t * gC);
    }
}

int fill_window(SsimWin& w) {
    // normalised fp32 Gaussian, sigma 1.5, taps -5..5
    float s = 0.f;
    for (int k = 0; k < SS_K; ++k) {
        const float d = (float)(k - SS_R) / 1.5f;
        w.g[k] = expf(-d * d / 2.f);
        s += w.g[k];
    }
    for (int k = 0; k < SS_K; ++k) w.g[k] /= s;
    return KSR_OK;
}

}  // namespace

extern "C" int64_t ksr_ssim_slab_floats(int C, int B, int H, int W) {
    if (C <= 0 || B <= 0 || H <= 2 * SS_R || W <= 2 * SS_R) return 0;
    const long tiles = (long)((H - 2 * SS_R + SS_T - 1) / SS_T) * ((W - 2 * SS_R + SS_T - 1) / SS_T);
    return tiles * C * B;
}

// ssim_out[0] = SSIM(pred, tgt) over C channels of B images (see the file header).  maps (optional):
// 3 * B * (H-10) * (W-10) floats receiving the per-window derivative maps of channel 0 for ksr_ssim_y_grad.
extern "C" int ksr_ssim_fwd(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride, int C,
                            int B, int H, int W, float* maps, float* slab, float* ssim_out, hipStream_t stream) {
    KSR_CHECK_ARG(pred && tgt && slab && ssim_out && C > 0 && B > 0, "ssim_fwd: bad argument");
    KSR_CHECK_ARG(H > 2 * SS_R && W > 2 * SS_R, "ssim_fwd: the image must be larger than the 11 x 11 window");
    KSR_CHECK_ARG(B <= 65535 && C <= 65535, "ssim_fwd: too many images / channels");
    SsimWin w;
    fill_window(w);
    SsimArgs a{pred, (long)pred_batch_stride, tgt, (long)tgt_batch_stride, maps, slab, C, B, H, W};
    const int Hi = H - 2 * SS_R, Wi = W - 2 * SS_R;
    const unsigned tiles = (unsigned)(((Hi + SS_T - 1) / SS_T) * ((Wi + SS_T - 1) / SS_T));
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tiles, C, B), dim3(256), 0, stream, a, w);
    KSR_LAUNCH_CHECK("ssim_fwd");
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(1024), 0, stream, slab, (long)tiles * C * B,
                       1.0 / ((double)B * C * Hi * Wi), ssim_out);
    KSR_LAUNCH_CHECK("ssim_fwd(finish)");
    return KSR_OK;
}

// grad[b, y, x] += scale * dSSIM_sum/dpred[b, 0, y, x], where SSIM_sum is the un-normalised sum of the window values
// of channel 0 (scale carries the loss weight and 1 / (B * C * (H-10) * (W-10))).  pred / tgt point at the luma planes.
extern "C" int ksr_ssim_y_grad(const float* maps, const float* pred, int64_t pred_batch_stride, const float* tgt,
                               int64_t tgt_batch_stride, float* grad, int64_t grad_batch_stride, int B, int H, int W,
                               float scale, hipStream_t stream) {
    KSR_CHECK_ARG(maps && pred && tgt && grad && B > 0 && B <= 65535, "ssim_y_grad: bad argument");
    KSR_CHECK_ARG(H > 2 * SS_R && W > 2 * SS_R, "ssim_y_grad: the image must be larger than the 11 x 11 window");
    SsimWin w;
    fill_window(w);
    SsimGradArgs a{maps, pred, (long)pred_batch_stride, tgt, (long)tgt_batch_stride, grad, (long)grad_batch_stride, B, H, W, scale};
    const unsigned tiles = (unsigned)(((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T));
    hipLaunchKernelGGL(ssim_grad_kernel, dim3(tiles, 1, B), dim3(256), 0, stream, a, w);
    KSR_LAUNCH_CHECK("ssim_y_grad");
    return KSR_OK;
}
