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

/* ---------------------------------------------------------------------------------------------
 * Hyper-encoder trunk (GazeTR.forward, conv_sep_kan/gazetr.py:100-119).  Activations between the
 * kernels are NHWC: [B*H*W, C] fp32 row-major, pixel index n = (b*H + y)*W + x.  Conv weights are
 * the PyTorch parameter tensors as they are ([C_out, C_in, kh, kw] contiguous).
 */

/* ffn1 (FFN 1 -> 1 -> 64: pointwise1, depthwise 3x3 pad 1, ReLU, pointwise2;
 * conv_sep_kan/unet_blocks.py:236-241 with in_features = 1) fused with ffn2.fn_1 (1x1 conv 64 -> 128
 * + ReLU; unet_blocks.py:256-259,268).  Reads the Y plane (plane 0) of the NCHW image.
 *   t_out [N,128]; save_v [N] (optional): the scalar relu(dw(pw1(Y))) per pixel for the backward. */
int ksr_stem_fwd(const float* x_img, int64_t img_batch_stride, const float* pw1_w, const float* pw1_b,
                 const float* dw_w, const float* dw_b, const float* pw2_w, const float* pw2_b,
                 const float* fn1_w, const float* fn1_b, float* t_out, float* save_v, int B, int H, int W,
                 ksr_stream_t stream);

/* 1x1 conv 64 -> n_out with activation act (0 none, 1 ReLU): FFN.pointwise1 (n_out 64, act 0;
 * unet_blocks.py:237) and GatedFFN.fn_1 (n_out 128, act 1; unet_blocks.py:268). in [N,64], out [N,n_out]. */
int ksr_pointwise_fwd(const float* in, const float* w, const float* bias, float* out, int n_out, int act,
                      int B, int H, int W, ksr_stream_t stream);

/* FFN tail: depthwise 3x3 (pad 1) + ReLU + pointwise2 (unet_blocks.py:238-240), 64 channels.
 * in [N,64] (output of pointwise1), out [N,64]; save_operand [N,64] optional = relu(dw(in)). */
int ksr_dwrelu_pw_fwd(const float* in, const float* dw_w, const float* dw_b, const float* w, const float* bias,
                      float* out, float* save_operand, int B, int H, int W, ksr_stream_t stream);

/* GatedFFN tail + the sigmoid GazeTR applies to it: t_in [N,128] = relu(fn_1(x)); value = t[:, :64],
 * gate = depthwise3x3(t[:, 64:]) ; out = sigmoid(fn_2(value * gate))  (unet_blocks.py:269-275,
 * gazetr.py:104,107).  save_gate [N,64] optional = the gate (after the depthwise conv). */
int ksr_gate_pw_fwd(const float* t_in, const float* gate_w, const float* gate_b, const float* w,
                    const float* bias, float* out, float* save_gate, int B, int H, int W, ksr_stream_t stream);

/* One TransformerEncoderLayer (conv_sep_kan/gazetr.py:53-66) applied the way GazeTR applies it
 * (gazetr.py:110-115): nn.MultiheadAttention(E=64, 2 heads, batch_first=True) on a tensor arranged
 * '(h w) b c', i.e. the attention sequence is the S = local-batch images at one pixel.
 *   in/out [S, HW, 64].  params[14]: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
 *   linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight,
 *   norm2.bias, and (optional, both or neither) the encoder's final norm.weight, norm.bias
 *   (TransformerEncoder.forward, gazetr.py:29-30) fused after the last layer.
 *   saves[10] (NULL for inference): qkv [N,192], ctx [N,64], xhat1 [N,64], rstd1 [N], s1 [N,64],
 *   hmid [N,32], xhat2 [N,64], rstd2 [N], xhat_final [N,64], rstd_final [N].
 *   p_drop > 0 enables the four dropout sites (counter-based generator keyed by seed). S <= 128. */
int ksr_batch_tx_layer_fwd(const float* in, float* out, const float* const* params, float* const* saves, int S,
                           int HW, float eps, float p_drop, uint64_t seed, ksr_stream_t stream);

/* feed (1x1 conv 64 -> P, gazetr.py:117) + ConvSepKanCell (conv_sep_kan_cells.py:136-156) + PixelShuffle
 * (conv_sep_kan.py:38) fused; the [N,P] coefficient tensor is never written.  in_dim = 1, out_dim =
 * upscale^2.  Writes plane 0 of out [B, C, rH, rW] (out_batch_stride elements between images).
 * save_us [N, O, 2] optional: (spline weight, spline sum) per output for the backward.
 * ksr_feed_kan_supported() tells whether a fused kernel exists for the configuration. */
int ksr_feed_kan_supported(int upscale, int n_coef, int spline_order);
int ksr_feed_kan_fwd(const float* feat, const float* x_img, int64_t img_batch_stride, const float* feed_w,
                     const float* feed_b, float* out, int64_t out_batch_stride, float* save_us, int B, int H,
                     int W, int upscale, int n_coef, const float* knots, int n_knots, int spline_order,
                     ksr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KSR_H */
