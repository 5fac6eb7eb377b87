/* libksr — C ABI of the MI355X-native ConvSepKAN hot path.
 *
 * Plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 * (the library never allocates, frees or retains user-visible memory).  All tensors are
 * fp32 unless stated.  Kernels are enqueued asynchronously on `stream`; no call
 * synchronises.  Return value: 0 on success, negative on error (KSR_ERR_*), message via
 * ksr_last_error() (thread-local).  No exception or abort crosses this boundary.
 *
 * Citations name the reference interface each entry point replaces
 * (paths relative to the reference repository).
 */
#ifndef KSR_H
#define KSR_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ksr_stream_t; /* == hipStream_t */

#define KSR_OK 0
#define KSR_ERR_ARG (-1)
#define KSR_ERR_UNSUPPORTED (-2)
#define KSR_ERR_HIP (-3)
#define KSR_ERR_COMM (-4)

int ksr_version(void);
const char* ksr_last_error(void);

/* KAN cell with the per-pixel parameter tensor materialised.
 * Replaces ConvSepKanCell.forward (conv_sep_kan/conv_sep_kan_cells.py:136-156), KANLayer.__call__
 * (conv_sep_kan/kan_layer.py:153-162), compute_bspline (conv_sep_kan/bspline.py:4-28) and, when
 * shuffle_r > 0, nn.PixelShuffle (conv_sep_kan/conv_sep_kan.py:38).
 *   x    [B, I, H, W]
 *   w    [B, P, H, W] (w_nhwc = 0) or [B*H*W, P] (w_nhwc = 1), P = O*I*(n_coef+2),
 *        per pixel: coef (o, i, g) | spline weight (o, i) | residual weight (o, i)
 *   out  [B, O, H, W], or with shuffle_r = r > 0: [B, O/r^2, rH, rW]; out_batch_stride (elements,
 *        0 = dense) lets the caller write the Y plane of a wider [B, 3, rH, rW] tensor.
 *   knots: HOST array of n_knots = n_coef + spline_order + 1 fp32 knots. spline_order <= 3. */
int ksr_kan_cell_fwd(const float* x, const float* w, float* out, int B, int I, int O, int H, int W,
                     int n_coef, const float* knots, int n_knots, int spline_order, int w_nhwc,
                     int shuffle_r, int64_t out_batch_stride, ksr_stream_t stream);

/* Backward of the above (autograd of the same reference lines): gw has w's layout and is fully
 * written; gx [B, I, H, W] may be NULL. gout has out's layout. */
int ksr_kan_cell_bwd(const float* x, const float* w, const float* gout, float* gw, float* gx, int B,
                     int I, int O, int H, int W, int n_coef, const float* knots, int n_knots,
                     int spline_order, int w_nhwc, int shuffle_r, int64_t gout_batch_stride,
                     ksr_stream_t stream);

/* Bilinear x r upsampling, align_corners = false, of n_channels planes starting at in_channel0,
 * written to planes out_channel0.. of out.  Replaces F.interpolate(..., mode='bilinear') on Cb/Cr
 * and the torch.concat that follows (conv_sep_kan/conv_sep_kan.py:30-34,40). Strides in elements. */
int ksr_bilinear_up(const float* in, float* out, int B, int n_channels, int H, int W, int r,
                    int64_t in_batch_stride, int in_channel0, int64_t out_batch_stride,
                    int out_channel0, ksr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KSR_H */
