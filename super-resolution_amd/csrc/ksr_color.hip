// RGB <-> YCbCr (SDTV / BT.601 range) on NCHW fp32 images: the steps either side of the model in the reference's
// data path (src/ml/utils/colors/ycbcr.py:13-76; dataset transform and GenerateCallback), plus the fused output
// stage of inference: bilinear Cb/Cr upsampling + YCbCr -> RGB in one pass over the output.
#include "ksr_common.h"

namespace {

struct ColorMat { float m[9]; float o[3]; };   // out[j] = sum_i in[i] * m[3*i + j] (+/- o, see kernels)

struct ColorArgs {
    const float* in; float* out;
    long npix;          // pixels per image plane
    int B;
    long in_bstride, out_bstride;
};

// ycbcr = rgb @ T + O   (ycbcr.py:38-39)
__global__ void __launch_bounds__(256) rgb_to_ycbcr_kernel(ColorArgs a, ColorMat c) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.npix;
    if (i >= total) return;
    const int b = (int)(i / a.npix);
    const long p = i - (long)b * a.npix;
    const float* s = a.in + (long)b * a.in_bstride + p;
    const float r = s[0], g = s[a.npix], bl = s[2 * a.npix];
    float* d = a.out + (long)b * a.out_bstride + p;
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j * a.npix] = fmaf(bl, c.m[6 + j], fmaf(g, c.m[3 + j], r * c.m[j])) + c.o[j];
}

// rgb = (ycbcr - O) @ Tinv   (ycbcr.py:70-71)
__global__ void __launch_bounds__(256) ycbcr_to_rgb_kernel(ColorArgs a, ColorMat c) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.npix;
    if (i >= total) return;
    const int b = (int)(i / a.npix);
    const long p = i - (long)b * a.npix;
    const float* s = a.in + (long)b * a.in_bstride + p;
    const float y = s[0] - c.o[0], cb = s[a.npix] - c.o[1], cr = s[2 * a.npix] - c.o[2];
    float* d = a.out + (long)b * a.out_bstride + p;
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j * a.npix] = fmaf(cr, c.m[6 + j], fmaf(cb, c.m[3 + j], y * c.m[j]));
}

struct UpRgbArgs {
    const float* x_lr;   // [B,3,H,W] YCbCr (planes 1, 2 are read)
    const float* y_hr;   // [B,rH,rW] upscaled luma
    float* out;          // [B,3,rH,rW] RGB
    int B, H, W, r;
    long y_bstride;
};

// out = ycbcr_to_rgb(concat(y_hr, bilinear(cb), bilinear(cr))): 4 output pixels of one row per thread
__global__ void __launch_bounds__(256) bilinear_up_to_rgb_kernel(UpRgbArgs a, ColorMat c) {
    const int OW = a.W * a.r, OH = a.H * a.r, OW4 = OW / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * OH * OW4;
    if (idx >= total) return;
    const int x4 = (int)(idx % OW4);
    const int oy = (int)((idx / OW4) % OH);
    const int b = (int)(idx / ((long)OW4 * OH));
    const float sc = 1.0f / (float)a.r;
    float sy = sc * ((float)oy + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
    const long HW = (long)a.H * a.W;
    const float* cbp = a.x_lr + ((long)b * 3 + 1) * HW;
    const float* crp = cbp + HW;
    const float4 yv = *reinterpret_cast<const float4*>(a.y_hr + (long)b * a.y_bstride + (long)oy * OW + x4 * 4);
    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
    float o[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ox = x4 * 4 + e;
        float sx = sc * ((float)ox + 0.5f) - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        const int x0 = (int)sx;
        const int x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
        const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
        const long i00 = (long)y0 * a.W + x0, i01 = (long)y0 * a.W + x1, i10 = (long)y1 * a.W + x0, i11 = (long)y1 * a.W + x1;
        const float cb = ly0 * (lx0 * cbp[i00] + lx1 * cbp[i01]) + ly1 * (lx0 * cbp[i10] + lx1 * cbp[i11]) - c.o[1];
        const float cr = ly0 * (lx0 * crp[i00] + lx1 * crp[i01]) + ly1 * (lx0 * crp[i10] + lx1 * crp[i11]) - c.o[2];
        const float y = yy[e] - c.o[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) o[j][e] = fmaf(cr, c.m[6 + j], fmaf(cb, c.m[3 + j], y * c.m[j]));
    }
    const long plane = (long)OH * OW;
    float* d = a.out + (long)b * 3 * plane + (long)oy * OW + x4 * 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(d + j * plane) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
}

int fill(ColorMat& c, const float* m9, const float* o3) {
    KSR_CHECK_ARG(m9 && o3, "colour conversion: matrix / offset is null");
    for (int i = 0; i < 9; ++i) c.m[i] = m9[i];
    for (int i = 0; i < 3; ++i) c.o[i] = o3[i];
    return KSR_OK;
}

}  // namespace

// ycbcr [B,3,H,W] = rgb @ T + O per pixel; m9 = T row-major ([in][out], host array), o3 = O (host array).
extern "C" int ksr_rgb_to_ycbcr(const float* rgb, float* ycbcr, int B, int H, int W, const float* m9, const float* o3,
                                hipStream_t stream) {
    KSR_CHECK_ARG(rgb && ycbcr && B > 0 && H > 0 && W > 0, "rgb_to_ycbcr: bad argument");
    ColorMat c;
    if (int rc = fill(c, m9, o3)) return rc;
    const long npix = (long)H * W;
    ColorArgs a{rgb, ycbcr, npix, B, 3 * npix, 3 * npix};
    hipLaunchKernelGGL(rgb_to_ycbcr_kernel, dim3((unsigned)((B * npix + 255) / 256)), dim3(256), 0, stream, a, c);
    KSR_LAUNCH_CHECK("rgb_to_ycbcr");
    return KSR_OK;
}

// rgb [B,3,H,W] = (ycbcr - O) @ Tinv per pixel; m9 = Tinv row-major (host array), o3 = O (host array).
extern "C" int ksr_ycbcr_to_rgb(const float* ycbcr, float* rgb, int B, int H, int W, const float* m9, const float* o3,
                                hipStream_t stream) {
    KSR_CHECK_ARG(rgb && ycbcr && B > 0 && H > 0 && W > 0, "ycbcr_to_rgb: bad argument");
    ColorMat c;
    if (int rc = fill(c, m9, o3)) return rc;
    const long npix = (long)H * W;
    ColorArgs a{ycbcr, rgb, npix, B, 3 * npix, 3 * npix};
    hipLaunchKernelGGL(ycbcr_to_rgb_kernel, dim3((unsigned)((B * npix + 255) / 256)), dim3(256), 0, stream, a, c);
    KSR_LAUNCH_CHECK("ycbcr_to_rgb");
    return KSR_OK;
}

// Output stage of RGB inference: out_rgb [B,3,rH,rW] = ycbcr_to_rgb(concat(y_hr, bilinear_up(Cb), bilinear_up(Cr)))
// with Cb, Cr = planes 1, 2 of x_lr [B,3,H,W] and y_hr [B,rH,rW] (images y_batch_stride elements apart).
// (W * r) must be a multiple of 4.
extern "C" int ksr_bilinear_up_to_rgb(const float* x_lr, const float* y_hr, int64_t y_batch_stride, float* out_rgb, int B,
                                      int H, int W, int r, const float* m9, const float* o3, hipStream_t stream) {
    KSR_CHECK_ARG(x_lr && y_hr && out_rgb && B > 0 && H > 0 && W > 0 && r >= 1, "bilinear_up_to_rgb: bad argument");
    KSR_CHECK_ARG((W * r) % 4 == 0 && y_batch_stride % 4 == 0, "bilinear_up_to_rgb: W*r and y_batch_stride must be multiples of 4");
    ColorMat c;
    if (int rc = fill(c, m9, o3)) return rc;
    UpRgbArgs a{x_lr, y_hr, out_rgb, B, H, W, r, (long)y_batch_stride};
    const long total = (long)B * (H * r) * (long)(W * r / 4);
    hipLaunchKernelGGL(bilinear_up_to_rgb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, c);
    KSR_LAUNCH_CHECK("bilinear_up_to_rgb");
    return KSR_OK;
}
