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

/* Precision mode of the kernels below.  The mode is a property of the CALLING THREAD (thread-local storage inside the
 * library), read by each entry point at launch time: ksr_set_precision changes it for the thread that calls it and for
 * no other, ksr_get_precision returns the calling thread's mode, and EVERY NEW THREAD STARTS IN MODE 0.  A program that
 * sets the mode on one thread and launches from another (a worker pool; autograd's backward thread) must set it again on
 * the launching thread: in mode 2 the activation buffers are bf16, and an entry point that runs in mode 0 would read them
 * as fp32 (twice the bytes: an out-of-bounds read, not an error code).  Concurrent callers with different modes do not
 * interfere.
 *   0  exact fp32 MFMA, fp32 storage: the mode in which parity with the reference's fp32 results is stated;
 *   1  bf16 MFMA operands with fp32 accumulation (1x1 convs / linears, attention products, weight gradients),
 *      fp32 storage (the reference has no such mode; separately stated tolerance);
 *   2  as 1, and every [N, C] activation / activation-gradient tensor exchanged between the kernels of the
 *      hyper-encoder path is STORED AS bf16: in this mode the `float*` parameters that carry such tensors
 *      point to bf16 buffers of the same shape.  Images, outputs, per-pixel scalars (v, rstd, dpre),
 *      parameters, parameter gradients and workspaces stay fp32.  The stand-alone KAN-cell, bilinear, loss,
 *      optimiser and UNet entry points have no mode-2 variant (fp32 tensors only); of the channel-attention entry points
 *      ksr_chan_dw_* and ksr_chan_apply_* take bf16 tensors in mode 2 (stated at their declarations), the per-image
 *      algebra (ksr_chan_soft_*) stays fp32.
 * The spline basis, LayerNorm, softmax and all accumulation are fp32 in every mode.
 * out_f32_copy / in_f32_copy arguments: the one tensor whose bf16 rounding matters (the sigmoid output that
 * enters the first transformer layer's residual + LayerNorm: values 0.5 +- 1e-3 at initialisation) can be kept
 * as an additional fp32 copy; NULL = not used. */
int ksr_set_precision(int prec);
int ksr_get_precision(void);

/* Mode 0 only, forward kernels that have such an instance (the single-image transformer layer of inference,
 * ksr_batch_tx_layer_fwd with S = 1): form the fp32 products on the bf16 matrix cores.  Both operands are split exactly
 * into three bf16 terms (24 significand bits), six of the nine partial products are accumulated in fp32 (the dropped ones
 * are below 2^-24 of a product): against a float64 evaluation the results are as close as those of the fp32 matrix
 * instruction (measured: max error 2.4e-7 against 2.8e-7 on outputs of magnitude 1), but not bit-equal to them.
 * v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate and adds its time to a kernel's vector instructions.
 * on = 1 / 0 for the calling thread, -1 = back to the process default (on, unless the environment has KSR_F32_SPLIT=0).
 * No counterpart in the reference, whose precision is whatever its torch build computes fp32 convolutions in. */
int ksr_set_f32_split(int on);
int ksr_get_f32_split(void);

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

/* ffn1, ffn2 and ffn3.pointwise1 in one pass over the Y plane (gazetr.py:102-105; FFN unet_blocks.py:236-241, GatedFFN
 * :267-275): ffn1's 64 channels are an affine function of one scalar v per pixel, so relu(fn_1(ffn1(Y))) is
 * relu(A v + B) per channel and the 128-channel tensor needs neither a GEMM nor a trip through memory.
 * params[14]: ffn1.pointwise1 {weight, bias}, ffn1.depthwise {weight [9], bias}, ffn1.pointwise2 {weight [64], bias},
 * ffn2.fn_1.0 {weight [128,64], bias}, ffn2.gate {weight [64,9], bias}, ffn2.fn_2 {weight [64,64], bias},
 * ffn3.pointwise1 {weight [64,64], bias}.  t3_out [N,64] = ffn3.pointwise1(sigmoid(ffn2(ffn1(Y)))).
 * Training passes v [N] (fp32), gate [N,64], x3 [N,64] = sigmoid(ffn2) and t [N,128] = relu(fn_1(x1)) and / or ab [256]
 * (fp32: A [128] | B [128]): the backward entry points with a `_v` suffix rebuild t from (v, ab) and do not need the
 * tensor.  Inference passes none. */
int ksr_stem_gate_fwd(const float* x_img, int64_t img_batch_stride, const float* const* params, float* t3_out,
                      float* save_v, float* save_t, float* save_gate, float* save_x3, float* save_ab, int B, int H, int W,
                      ksr_stream_t stream);

/* 1x1 conv 64 -> n_out with activation act (0 none, 1 ReLU): FFN.pointwise1 (n_out 64, act 0;
 * unet_blocks.py:237), GatedFFN.fn_1 (n_out 128, act 1; unet_blocks.py:268) and the attention in_proj
 * (n_out 192, act 0; nn.MultiheadAttention in gazetr.py:39). in [N,64], out [N,n_out]. */
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
                    const float* bias, float* out, float* save_gate, float* out_f32_copy, int B, int H, int W,
                    ksr_stream_t stream);

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
                           int HW, float eps, float p_drop, uint64_t seed, const float* in_f32_copy,
                           ksr_stream_t stream);

/* Both encoder layers and the final LayerNorm of ONE image in one launch (GazeTR.forward in eval mode, gazetr.py:110-117, on a
 * batch of one: the attention over a single key is the V projection, so each layer applies W_out W_v as one matrix).  The
 * [N,64] tensor between the layers is never written.  params0 / params1: the 12 per-layer tensors in the order of
 * ksr_batch_tx_layer_fwd's params; in, out [N,64] fp32.  Mode 0 only (KSR_ERR_UNSUPPORTED otherwise); follows
 * ksr_set_f32_split. */
int ksr_tx2_single_fwd(const float* in, float* out, const float* const* params0, const float* const* params1,
                       const float* nf_w, const float* nf_b, int64_t N, float eps, ksr_stream_t stream);

/* The same layer as three kernels, used for large local batches: ksr_pointwise_fwd(n_out = 192) computes
 * qkv [N,192] = in_proj(x); ksr_batch_attn_fwd runs the 2-head softmax attention over the S images of every
 * pixel on the matrix cores (ctx [N,64]); ksr_batch_tx_layer_post_fwd does out_proj, norm1, the MLP, norm2
 * (and the final norm).  params / saves as for ksr_batch_tx_layer_fwd (params[0..1], saves[0..1] unused). */
int ksr_batch_attn_fwd(const float* qkv, float* ctx, int S, int HW, float p_drop, uint64_t seed,
                       ksr_stream_t stream);
/* ksr_pointwise_fwd(n_out = 192) + ksr_batch_attn_fwd in one launch (MultiheadAttention's in_proj and attention,
 * gazetr.py:110-115 through torch's multi_head_attention_forward): qkv [N,192] is written when not NULL (the backward
 * reads it) and never read.  The fused kernel exists for bf16 storage with 3 <= S <= 64; other cases run the two
 * kernels and need the qkv buffer. */
int ksr_batch_qkv_attn_fwd(const float* in, const float* in_proj_weight, const float* in_proj_bias, float* qkv,
                           float* ctx, int S, int HW, float p_drop, uint64_t seed, ksr_stream_t stream);
int ksr_batch_tx_layer_post_fwd(const float* in, const float* ctx, float* out, const float* const* params,
                                float* const* saves, int64_t N, float eps, float p_drop, uint64_t seed,
                                const float* in_f32_copy, ksr_stream_t stream);

/* feed (1x1 conv 64 -> P, gazetr.py:117) + ConvSepKanCell (conv_sep_kan_cells.py:136-156) + PixelShuffle
 * (conv_sep_kan.py:38) fused; the [N,P] coefficient tensor is never written.  in_dim = 1, out_dim =
 * upscale^2.  Writes plane 0 of out [B, C, rH, rW] (out_batch_stride elements between images).
 * save_us [N, O, 2] optional: (spline weight, spline sum) per output for the backward.
 * ksr_feed_kan_supported() tells whether a fused kernel exists for the configuration: spline_order 3,
 * upscale 2 or 4, n_coef = grid_size + spline_order in {6, 8, 11, 13, 19}; spline_order 1 / 2 with grid_size 5. */
int ksr_feed_kan_supported(int upscale, int n_coef, int spline_order);
int ksr_feed_kan_fwd(const float* feat, const float* x_img, int64_t img_batch_stride, const float* feed_w,
                     const float* feed_b, float* out, int64_t out_batch_stride, float* save_us, int B, int H,
                     int W, int upscale, int n_coef, const float* knots, int n_knots, int spline_order,
                     ksr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward.  The reference has no hand-written backward: these replace what autograd derives from the
 * forward lines cited above.  "saved" tensors are the optional outputs of the forward entry points.
 */

/* Backward of ksr_feed_kan_fwd.  gout: gradient of the Y plane of the output (pixel-shuffled layout,
 * gout_batch_stride elements between images).  Writes gfeat [N,64] and gparam [N, 2, L] = gradient of
 * the feed output in the kernel's lane order, L = ksr_feed_kan_lane_len(); feed.weight/bias gradients
 * follow from ksr_wgrad(gparam, feat, ..., kan_upscale, kan_n_coef). */
int ksr_feed_kan_lane_len(int upscale, int n_coef);
int ksr_feed_kan_bwd(const float* gout, int64_t gout_batch_stride, const float* x_img, int64_t img_batch_stride,
                     const float* feed_w, const float* save_us, float* gfeat, float* gparam, int B, int H, int W,
                     int upscale, int n_coef, const float* knots, int n_knots, int spline_order,
                     ksr_stream_t stream);

/* Backward of ksr_batch_tx_layer_fwd, per-token part: [final norm] <- norm2 <- linear2/ReLU/linear1 <-
 * norm1 <- out_proj.  params[6]: final norm.weight (or NULL), norm2.weight, norm1.weight, linear2.weight,
 * linear1.weight, out_proj.weight.  saved[7]: xhat_final, rstd_final, xhat2, rstd2, hmid, xhat1, rstd1.
 * outs[7]: ds2 [N,64] (gradient of norm2's output; only with the final norm), dm [N,64] (of linear2's
 * output), dpre [N,32] (of linear1's output), ds1 [N,64] (of norm1's output), du1 [N,64] (of norm1's
 * input), da [N,64] (of out_proj's output; only when p_drop > 0, else it equals du1), dctx [N,64].
 * ln_slab (optional, 3 * 768 * 128 floats of workspace): the kernel then also reduces the LayerNorm gain /
 * shift gradients; ln_grads[3] (final norm, norm2, norm1; NULL entries are skipped) each receive the gain
 * gradient at [c] and the shift gradient at [ln_row_stride + c], c < 64, and ds2 / ds1, which only feed
 * those sums (ksr_colreduce otherwise), may be NULL. */
int ksr_batch_tx_layer_bwd_token(const float* dy, const float* const* params, const float* const* saved,
                                 float* const* outs, int64_t N, float p_drop, uint64_t seed, float* ln_slab,
                                 float* const* ln_grads, int ln_row_stride, ksr_stream_t stream);

/* Attention backward over the S images of each pixel: qkv [N,192], ctx [N,64] (saved), dctx -> dqkv [N,192]. */
int ksr_batch_tx_layer_bwd_attn(const float* qkv, const float* ctx, const float* dctx, float* dqkv, int S, int HW,
                                float p_drop, uint64_t seed, ksr_stream_t stream);

/* The same computation on the matrix cores (used for large S, the backward of ksr_batch_attn_fwd). */
int ksr_batch_attn_bwd(const float* qkv, const float* ctx, const float* dctx, float* dqkv, int S, int HW,
                       float p_drop, uint64_t seed, ksr_stream_t stream);

/* ds = resid + dqkv * in_proj_weight; with y_saved != NULL also times y(1-y) (the sigmoid that produced
 * the first transformer layer's input, gazetr.py:107). */
int ksr_in_proj_bwd(const float* dqkv, const float* in_w, const float* resid, const float* y_saved, float* ds, int B,
                    int H, int W, ksr_stream_t stream);

/* ksr_in_proj_bwd and the in-projection's weight / bias gradient (ksr_wgrad(dqkv, x_in)) in one pass over dqkv, the widest
 * tensor of the transformer backward: bf16 storage mode (ksr_set_precision(2)) only, other modes return
 * KSR_ERR_UNSUPPORTED.  x_in [N,64] is the layer's input; sigmoid_input != 0: ds is also multiplied by x_in (1 - x_in) (the
 * first layer, whose input is a sigmoid output, gazetr.py:107).  slab: ksr_in_proj_bwd_wg_slab_floats() floats of
 * workspace; dW [192,64], db [192].  Deterministic. */
int64_t ksr_in_proj_bwd_wg_slab_floats(void);
int ksr_in_proj_bwd_wg(const float* dqkv, const float* in_w, const float* resid, const float* x_in, int sigmoid_input,
                       float* ds, float* slab, float* dW, float* db, int64_t N, ksr_stream_t stream);

/* GatedFFN tail backward (backward of ksr_gate_pw_fwd without its sigmoid, which the producer of dz has
 * already applied): dA = dz * fn_2.weight; dt_value = dA * gate * (t[:, :64] > 0) written to channels 0..63
 * of dt [N,128]; dgate [N,64] = dA * t[:, :64]. */
int ksr_gate_pw_bwd(const float* dz, const float* fn2_w, const float* gate_saved, const float* t_saved,
                    float* dt_value, float* dgate, int B, int H, int W, ksr_stream_t stream);

/* fn_1 (1x1 conv 64 -> 128 + ReLU) backward.  Completes dt[:, 64:128] = convT3x3(dgate) * (t[:, 64:] > 0),
 * then dx [N,64] = dt * fn_1.weight.  stem = 1: the conv's input was ffn1 (ksr_stem_fwd); additionally
 * dpre [N] = (sum_c dx[c] * pointwise2.weight[c]) * (v > 0), the gradient at ffn1's depthwise output.
 * stem with sums_slab (3 * 512 * 128 floats of workspace, else NULL): the kernel also reduces over all pixels
 * dt_sums[0][c] = sum dt[n,c] v[n], dt_sums[1][c] = sum dt[n,c] (c < 128; rows 128 floats apart) and
 * dx_sums[c] = sum dx[n,c] v[n], dx_sums[dx_row_stride + c] = sum dx[n,c] (c < 64) - everything the stem's
 * parameter gradients need of dt[:, 64:128] and dx, which are then not written (dx may be NULL). */
int ksr_fn1_bwd(float* dt, const float* dgate, const float* gate_w, const float* t_saved, const float* fn1_w,
                float* dx, int stem, const float* v_saved, const float* pw2_w, float* dpre, float* sums_slab,
                float* dt_sums, float* dx_sums, int dx_row_stride, int B, int H, int W, ksr_stream_t stream);

/* FFN.pointwise2 backward: ds [N,64] = (dy * pointwise2.weight) * (operand_saved > 0). */
int ksr_pw2_bwd(const float* dy, const float* pw2_w, const float* operand_saved, float* ds, int B, int H, int W,
                ksr_stream_t stream);

/* FFN.depthwise + pointwise1 backward: dt [N,64] = convT3x3(ds); dx = dt * pointwise1.weight;
 * dz_prev [N,64] = dx * y(1-y), y_saved = the sigmoid output that fed the FFN (gazetr.py:104). */
int ksr_pw1_bwd(const float* ds, const float* dw_w, const float* pw1_w, const float* y_saved, float* dt,
                float* dz_prev, int B, int H, int W, ksr_stream_t stream);

/* ksr_gate_pw_bwd / ksr_pw2_bwd with the weight and bias gradient of their 64 -> 64 conv (fn_2 with input value * gate;
 * pointwise2 with input operand_saved) accumulated in the same pass: both operands of that product are already in the
 * kernel's registers, so the separate ksr_wgrad launch and its second read of them are saved.  bf16 storage mode
 * (ksr_set_precision(2)) only; other modes return KSR_ERR_UNSUPPORTED.  slab: ksr_fused_wgrad_slab_floats() floats of
 * workspace; dW [64,64], db [64].  Deterministic (fixed-order reductions). */
int64_t ksr_fused_wgrad_slab_floats(void);
int ksr_gate_pw_bwd_wg(const float* dz, const float* fn2_w, const float* gate_saved, const float* t_saved,
                       float* dt_value, float* dgate, float* slab, float* dW, float* db, int B, int H, int W,
                       ksr_stream_t stream);
int ksr_pw2_bwd_wg(const float* dy, const float* pw2_w, const float* operand_saved, float* ds, float* slab, float* dW,
                   float* db, int B, int H, int W, ksr_stream_t stream);
/* ksr_pw1_bwd with ffn3.pointwise1's weight / bias gradient (dt^T y_saved) accumulated in the same pass (bf16 storage mode;
 * the four 32 x 32 blocks of the product are split over the workgroup's four waves): dt, which only fed that product, is
 * not written.  slab: ksr_fused_wgrad_slab_floats() floats. */
int ksr_pw1_bwd_wg(const float* ds, const float* dw_w, const float* pw1_w, const float* y_saved, float* dz_prev, float* slab,
                   float* dW, float* db, int B, int H, int W, ksr_stream_t stream);

/* First gated block (ffn2, whose input is ffn1): variants of ksr_gate_pw_bwd(_wg), ksr_fn1_bwd (stem form with column sums)
 * and ksr_dwconv_wgrad that rebuild the saved tensor t [N,128] = relu(fn_1(ffn1(Y))) from the per-pixel scalar v_saved [N]
 * and ab = A [128] | B [128] (both written by ksr_stem_gate_fwd): t[n][c] = relu(A[c] v[n] + B[c]).  The tensor is then
 * neither written by the forward nor read by the backward.  ksr_gate_pw_bwd_v: slab/dW/db non-NULL (bf16 storage mode)
 * also accumulates fn_2's weight gradient as ksr_gate_pw_bwd_wg does; NULL = data gradient only, any mode. */
int ksr_gate_pw_bwd_v(const float* dz, const float* fn2_w, const float* gate_saved, const float* v_saved, const float* ab,
                      float* dt_value, float* dgate, float* slab, float* dW, float* db, int B, int H, int W,
                      ksr_stream_t stream);
int ksr_fn1_bwd_stem_v(float* dt, const float* dgate, const float* gate_w, const float* v_saved, const float* ab,
                       const float* fn1_w, const float* pw2_w, float* dpre, float* sums_slab, float* dt_sums, float* dx_sums,
                       int dx_row_stride, int B, int H, int W, ksr_stream_t stream);
int ksr_dwconv_wgrad_v(const float* v_saved, const float* ab, const float* ds, int ds_ld, int ds_c0, float* slab, float* dW,
                       float* db, int B, int H, int W, ksr_stream_t stream);

/* Weight gradient of a 1x1 conv / linear layer: dW[j][k] = sum_n dz[n, dz_c0+j] * a1[n, a1_c0+k]
 * (* a2[n, a2_c0+k] when a2 != NULL), db[j] = sum_n dz[n, dz_c0+j].  J % 32 == 0, K in {32, 64}.
 * slab: ksr_wgrad_slab_floats(J, K) floats of workspace.  kan_upscale > 0: dz is ksr_feed_kan_bwd's
 * gparam and rows are scattered to the reference's channel order of feed.weight / feed.bias.
 * Deterministic (fixed-order slab reduction, no atomics). */
int ksr_wgrad_slab_floats(int J, int K);
int ksr_wgrad(const float* dz, int dz_ld, int dz_c0, int J, const float* a1, int a1_ld, int a1_c0, int K,
              const float* a2, int a2_ld, int a2_c0, int64_t N, float* slab, float* dW, float* db,
              int kan_upscale, int kan_n_coef, ksr_stream_t stream);

/* out[0][c] = sum_n a[n, a_c0+c] * b, out[1][c] = sum_n a[n, a_c0+c], c < 64; b is [N,64] or, with
 * b_is_per_pixel_scalar, [N].  LayerNorm gain/shift gradients and ffn1.pointwise2.  slab: 1024*128 floats. */
int ksr_colreduce(const float* a, int a_ld, int a_c0, const float* b, int b_is_per_pixel_scalar, int64_t N,
                  float* slab, float* out, int out_row_stride, ksr_stream_t stream);

/* Depthwise 3x3 parameter gradients (64 channels): in = the conv's input (NHWC, in_ld channels per pixel,
 * 64 channels from in_c0), ds = gradient of its output (ds_ld channels per pixel, 64 from ds_c0).
 * dW [64,1,3,3], db [64]. slab: 512*640 floats. */
int ksr_dwconv_wgrad(const float* in, int in_ld, int in_c0, const float* ds, int ds_ld, int ds_c0, float* slab,
                     float* dW, float* db, int B, int H, int W, ksr_stream_t stream);
/* The same for `groups` (1..8) depthwise convs over consecutive blocks of 64 channels of the same two tensors in ONE launch:
 * the q_dw | k_dw | v_dw of the channel-attention head (conv_sep_kan_attention.py:16-37) under autograd.  dW [groups*64][9],
 * db [groups*64]; slab: groups * 512 * 640 floats. */
int ksr_dwconv_wgrad_groups(const float* in, int in_ld, int in_c0, const float* ds, int ds_ld, int ds_c0, int groups,
                            float* slab, float* dW, float* db, int B, int H, int W, ksr_stream_t stream);

/* ffn1's scalar front (pointwise1 1->1, depthwise 3x3 on one channel): out12 = gradients of
 * pointwise1.weight, pointwise1.bias, depthwise.weight (9), depthwise.bias.  slab: 256*12 floats. */
int ksr_stem_scalar_wgrad(const float* x_img, int64_t img_batch_stride, const float* dpre, const float* pw1_w,
                          const float* pw1_b, const float* dw_w, float* slab, float* out12, int B, int H, int W,
                          ksr_stream_t stream);

/* Plain 1x1-conv data gradient dx [N,64] = dz [N,n_out] * W, W [n_out][64], n_out 64 or 192 (backward of
 * ksr_pointwise_fwd without activation). */
int ksr_pointwise_bwd(const float* dz, int n_out, const float* w, float* dx, int B, int H, int W, ksr_stream_t stream);

/* bf16 storage mode: the out-projection's backward continued from du1 (the gradient that ksr_batch_tx_layer_bwd_token
 * leaves when its dctx output is NULL): dctx [N,64] = mask(du1) * W_out, with the attention branch's dropout mask
 * (gazetr.py:55-56 at p = 0.1) regenerated from seed, and dW [64][64] / db [64] of self_attn.out_proj from mask(du1)
 * and the saved ctx in the same pass.  wg_slab: ksr_tx_bwd_wg_slab_floats() floats of workspace. */
int64_t ksr_tx_bwd_wg_slab_floats(void);
int ksr_batch_tx_layer_bwd_out_wg(const float* du1, const float* ctx, const float* out_w, float* dctx, int64_t N,
                                  float p_drop, uint64_t seed, float* wg_slab, float* dW, float* db, ksr_stream_t stream);

/* bf16-storage mode: the per-token backward of the layer (the part of ksr_batch_tx_layer_bwd_token that stops at du1) as two
 * kernels that each accumulate one of the MLP's weight gradients from operands they hold, replacing that call and the two
 * ksr_wgrad launches for linear2 / linear1 (TransformerEncoderLayer.forward under autograd, gazetr.py:53-66).
 * ..._mlp_wg: norm2 <- linear2 <- ReLU / dropout.  dy [N,64]: gradient of the layer's output (for the last layer: of the
 *   final norm's input, see ksr_feed_kan_bwd_lnf); xhat2, rstd2, hmid: saved by the forward.  Writes du2 [N,64] (gradient
 *   entering the second residual branch) and dpre [N,32]; ln_grad: norm2's gain gradient at [c], shift gradient at
 *   [ln_row_stride + c]; dW2 [64,32], db2 [64].  ln_slab: 768 * 128 floats; wg_slab: ksr_tx_bwd_wg_slab_floats().
 * ..._ln1_wg: du1 [N,64] = LayerNorm-1 backward of du2 + dpre W1; ln_grad: norm1's (gain, shift) gradient pair; dW1 [32,64]
 *   is the product with xhat1 (map it through norm1 with ksr_wgrad_affine_input), db1 [32].  ln_slab: 768 * 128 floats. */
int ksr_batch_tx_layer_bwd_mlp_wg(const float* dy, const float* xhat2, const float* rstd2, const float* hmid,
                                  const float* n2_w, const float* l2_w, float* du2, float* dpre, int64_t N, float p_drop,
                                  uint64_t seed, float* ln_slab, float* ln_grad, int ln_row_stride, float* wg_slab,
                                  float* dW2, float* db2, ksr_stream_t stream);
int ksr_batch_tx_layer_bwd_ln1_wg(const float* du2, const float* dpre, const float* xhat1, const float* rstd1,
                                  const float* n1_w, const float* l1_w, float* du1, int64_t N, float* ln_slab,
                                  float* ln_grad, int ln_row_stride, float* wg_slab, float* dW1, float* db1,
                                  ksr_stream_t stream);

/* Deferred slab reductions (no counterpart in the reference, whose parameter gradients come from autograd:
 * src/ml/pipelines/default.py:75-84).  Every entry point that accumulates a parameter gradient writes per-workgroup
 * partial rows into its slab and ends with a small fixed-order reduction launch.  After ksr_reduce_defer(1) the CALLING
 * THREAD's entry points only record that reduction (up to 30; a full list is flushed on the stream of the call that
 * overflows it) and ksr_reduce_flush(stream) runs every recorded one in a single launch: same sums, same order, same
 * bits as the immediate form.  While the mode is on the caller gives every such call its OWN slab, keeps slabs and
 * outputs alive and untouched until the flush, and orders the flush after every producer (same stream, or an event).
 * Reductions over at most 32 partial rows and the batched (per-image) ones are never deferred.
 * Recorded reductions are kept in one list per stream of the recording calls (at most 4 streams per thread).
 * ksr_reduce_defer(k), k >= 2: as 1, and a stream's list is run on that stream as soon as it holds k entries (fewer
 * launches while the slabs are still in the caches; the per-call slabs are needed all the same).
 * ksr_reduce_defer(0) ends the mode (error if reductions are still recorded), ksr_reduce_defer(-1) ends it and drops what
 * was recorded (a caller's error path); ksr_reduce_pending() = recorded count. */
int ksr_reduce_defer(int on);
int ksr_reduce_flush(ksr_stream_t stream);
int ksr_reduce_pending(void);

/* Weight gradient of a linear map fed by a LayerNorm output s = xhat * gain + shift (linear1 after norm1,
 * gazetr.py:56-58) when the product was taken with xhat instead of s, which then need not be kept for the backward:
 * dW[j][k] <- dW[j][k] * gain[k] + db[j] * shift[k], in place. */
int ksr_wgrad_affine_input(float* dW, const float* db, const float* gain, const float* shift, int J, int K,
                           ksr_stream_t stream);

/* out[j][k] = r[j]*u[k] + c[j]*w[k] (weight gradient of ffn2.fn_1, whose input is u*v+w with one scalar v
 * per pixel); c_copy (optional) receives c, the bias gradient. */
int ksr_outer2(float* out, const float* r, const float* u, const float* c, const float* w, int J, int K,
               float* c_copy, ksr_stream_t stream);

/* SSIM term of the reference's training loss (src/ml/pipelines/default.py:79-81: L1(Y) + 0.15 (1 - SSIM)) with
 * torchmetrics' defaults for SSIM(data_range=(0, 1)) restated from the published definition (torchmetrics is not
 * in the reference tree: parity unpinned): inputs clamped to [0,1], 11x11 Gaussian window sigma 1.5, c1 = 0.01^2,
 * c2 = 0.03^2, variances clamped at 0, only windows entirely inside the image, mean over pixels, channels, images.
 * maps (optional, 3*B*(H-10)*(W-10) floats) receives the per-window derivatives of channel 0 for ksr_ssim_y_grad.
 * slab: ksr_ssim_slab_floats(C, B, H, W) floats. */
int64_t ksr_ssim_slab_floats(int C, int B, int H, int W);
int ksr_ssim_fwd(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride, int C, int B,
                 int H, int W, float* maps, float* slab, float* ssim_out, ksr_stream_t stream);
/* grad[b, y, x] += scale * d(sum of the channel-0 window values)/d pred[b, 0, y, x]; pred / tgt point at the luma
 * planes; scale carries the loss weight and the 1 / (B*C*(H-10)*(W-10)) of the mean. */
int ksr_ssim_y_grad(const float* maps, const float* pred, int64_t pred_batch_stride, const float* tgt,
                    int64_t tgt_batch_stride, float* grad, int64_t grad_batch_stride, int B, int H, int W, float scale,
                    ksr_stream_t stream);
/* Both terms of the reference's loss gradient (src/ml/pipelines/default.py:79-81) in one pass over the luma planes:
 * grad[b, y, x] = l1_grad_scale * sign(pred - tgt) / (B*H*W) + scale * d(SSIM sum)/d pred (written, not accumulated)
 * and l1_out[0] = mean |pred[:, 0] - tgt[:, 0]|: ksr_l1_y_loss followed by ksr_ssim_y_grad without the second read of
 * the planes.  l1_slab: ksr_ssim_l1_slab_floats(B, H, W) floats of workspace.  loss_out (optional; ssim_value is then the
 * device scalar ksr_ssim_fwd wrote): loss_out[0] = l1_out[0] + ssim_weight * (1 - ssim_value[0]), the loss of
 * default.py:79-81, each operation rounded to fp32 on its own. */
int64_t ksr_ssim_l1_slab_floats(int B, int H, int W);
int ksr_ssim_y_grad_l1(const float* maps, const float* pred, int64_t pred_batch_stride, const float* tgt,
                       int64_t tgt_batch_stride, float* grad, int64_t grad_batch_stride, int B, int H, int W, float scale,
                       float l1_grad_scale, float* l1_slab, float* l1_out, const float* ssim_value, float ssim_weight,
                       float* loss_out, ksr_stream_t stream);

/* ksr_feed_kan_bwd with the backward of the LayerNorm that produced feat - the encoder's final norm (gazetr.py:113, under
 * autograd) - inside (bf16-storage mode only): gfeat receives the gradient of that LayerNorm's INPUT (the last transformer
 * layer's output), ln_grad the norm's gain gradient at [c] and shift gradient at [ln_row_stride + c].  xhat [N,64], rstd [N]:
 * as saved by ksr_batch_tx_layer_post_fwd; ln_slab: 768 * 128 floats of workspace. */
int ksr_feed_kan_bwd_lnf(const float* gout, int64_t gout_batch_stride, const float* x_img, int64_t img_batch_stride,
                         const float* feed_w, const float* save_us, float* gfeat, float* gparam, int B, int H, int W,
                         int upscale, int n_coef, const float* knots, int n_knots, int spline_order, const float* xhat,
                         const float* rstd, const float* ln_w, float* ln_slab, float* ln_grad, int ln_row_stride,
                         ksr_stream_t stream);

/* The fused feed + KAN kernels with ONE WEIGHT MATRIX PER IMAGE (feed_w_per_image [B][P][64], P = r^2 (n_coef + 2); the
 * bias [P] is shared) and the pixel's 64 features at channels feat_c0 .. feat_c0 + 63 of rows of feat_ld elements.  This
 * is the channel-attention head of BASELINE configs[3] feeding the KAN cell: ConvSepKanAttention's last step,
 * a_proj(softmax(q^ k^T) v) (conv_sep_kan_attention.py:62-69), is linear in v with the per-image matrix
 * M_b = W_a softmax_b, so the [N, P] tensor of per-pixel parameters (ConvSepKanCell's `w`, conv_sep_kan_cells.py:136-156)
 * is never formed.  Otherwise as ksr_feed_kan_fwd / ksr_feed_kan_bwd; the backward writes the feature gradient to
 * channels gfeat_c0 .. of rows of gfeat_ld elements.  ksr_gram_batched_kan turns gparam (lane order, J =
 * 2 * ksr_feed_kan_lane_len) into the per-image dM_b [B][P][K] and the column sums tail [B][J] (first P entries of each
 * row, channel order): ksr_gram_batched with the rows scattered back to the reference's channel numbering. */
int ksr_feed_kan_fwd_img(const float* feat, int feat_ld, int feat_c0, const float* x_img, int64_t img_batch_stride,
                         const float* feed_w_per_image, const float* feed_b, float* out, int64_t out_batch_stride,
                         float* save_us, int B, int H, int W, int upscale, int n_coef, const float* knots, int n_knots,
                         int spline_order, ksr_stream_t stream);
int ksr_feed_kan_bwd_img(const float* gout, int64_t gout_batch_stride, const float* x_img, int64_t img_batch_stride,
                         const float* feed_w_per_image, const float* save_us, float* gfeat, int gfeat_ld, int gfeat_c0,
                         float* gparam, int B, int H, int W, int upscale, int n_coef, const float* knots, int n_knots,
                         int spline_order, ksr_stream_t stream);
int ksr_gram_batched_kan(const float* gparam, int J, const float* a1, int a1_ld, int a1_c0, int K, int64_t tokens_per_entry,
                         int nb, float* slab, float* out, float* tail, int kan_upscale, int kan_n_coef, ksr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient with respect to the INPUT IMAGE (what the reference's autograd returns when x requires a gradient; its
 * training step never asks for it).  Three contributions, each an entry point:
 *  - Cb / Cr planes: the adjoint of F.interpolate(..., mode='bilinear') (conv_sep_kan.py:33-34): ksr_bilinear_up_bwd,
 *    arguments as ksr_bilinear_up with gout in the place of out and gin (written) in the place of in;
 *  - luma, through the KAN evaluation's own dependence on the pixel value (kan_layer.py:68-84,106-118, bspline.py:4-28,
 *    the order-0 indicator held constant as autograd does): ksr_feed_kan_dx, gx written;
 *  - luma, through the encoder (ffn1's front, unet_blocks.py:236-238): ksr_stem_dx adds w1 * convT(dpre) to gx, where
 *    dpre [N] is the output of ksr_fn1_bwd(stem = 1) / ksr_fn1_bwd_stem_v. */
int ksr_bilinear_up_bwd(const float* gout, float* gin, int B, int n_channels, int H, int W, int r, int64_t in_batch_stride,
                        int in_channel0, int64_t out_batch_stride, int out_channel0, ksr_stream_t stream);
int ksr_feed_kan_dx(const float* feat, const float* x_img, int64_t img_batch_stride, const float* feed_w,
                    const float* feed_b, const float* gout, int64_t gout_batch_stride, float* gx, int64_t gx_batch_stride,
                    int B, int H, int W, int upscale, int n_coef, const float* knots, int n_knots, int spline_order,
                    ksr_stream_t stream);
int ksr_stem_dx(const float* dpre, const float* pw1_w, const float* dw_w, float* gx, int64_t gx_batch_stride, int B, int H,
                int W, ksr_stream_t stream);
/* ksr_feed_kan_dx behind the channel-attention head (conv_sep_kan_attention.py:62-69): the per-image weights and feature row
 * layout of ksr_feed_kan_fwd_img. */
int ksr_feed_kan_dx_img(const float* feat, int feat_ld, int feat_c0, const float* x_img, int64_t img_batch_stride,
                        const float* feed_w_per_image, const float* feed_b, const float* gout, int64_t gout_batch_stride,
                        float* gx, int64_t gx_batch_stride, int B, int H, int W, int upscale, int n_coef, const float* knots,
                        int n_knots, int spline_order, ksr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Colour pre/post-processing: the steps either side of the model in the reference's data path
 * (src/ml/utils/colors/ycbcr.py; dataset transform src/ml/datasets/huawei/datamodule.py:113-134 and
 * GenerateCallback src/ml/callbacks/generate.py:43,65,80).  NCHW fp32 images; m9 / o3 are HOST arrays.
 * ------------------------------------------------------------------------------------------- */

/* rgb_to_ycbcr (ycbcr.py:13-41): ycbcr = rgb @ T + O per pixel; m9 = T row-major ([in][out]), o3 = O. */
int ksr_rgb_to_ycbcr(const float* rgb, float* ycbcr, int B, int H, int W, const float* m9, const float* o3,
                     ksr_stream_t stream);
/* ycbcr_to_rgb (ycbcr.py:44-73): rgb = (ycbcr - O) @ Tinv per pixel; m9 = Tinv row-major, o3 = O. */
int ksr_ycbcr_to_rgb(const float* ycbcr, float* rgb, int B, int H, int W, const float* m9, const float* o3,
                     ksr_stream_t stream);
/* Fused output stage of RGB inference: out_rgb [B,3,rH,rW] = ycbcr_to_rgb(concat(y_hr, up(Cb), up(Cr))), i.e.
 * F.interpolate(bilinear) + torch.concat (conv_sep_kan.py:30-34,40) followed by ycbcr_to_rgb (generate.py:80)
 * in one pass; y_hr [B,rH,rW] with images y_batch_stride elements apart; W*r a multiple of 4. */
int ksr_bilinear_up_to_rgb(const float* x_lr, const float* y_hr, int64_t y_batch_stride, float* out_rgb, int B, int H,
                           int W, int r, const float* m9, const float* o3, ksr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Channel attention: ConvSepKanAttention.forward (conv_sep_kan/conv_sep_kan_attention.py:48-71) with
 * in = mid = 64 channels on NHWC pixels.  q | k | v live in one [N,192] tensor.  All 64 x 64 per-image
 * algebra (normalisation, softmax, a_proj) is folded into per-image matrices so that the attention output
 * and its gradient are never materialised:  y[n] = (W_a A_b) v[n] + b_a.
 * Forward:  ksr_pointwise_fwd(n_out=192) -> ksr_chan_dw_fwd -> ksr_gram_batched(q, k, want_sq) ->
 *           ksr_chan_soft_fwd -> ksr_chan_apply(v, M).
 * Backward: ksr_gram_batched(dy, v) -> ksr_chan_soft_bwd -> ksr_chan_apply(dy, M^T) / ksr_chan_apply([q|k], Mqk)
 *           -> ksr_chan_dw_bwd -> ksr_dwconv_wgrad x3, ksr_wgrad, ksr_pointwise_bwd (bf16 storage: ksr_chan_dw_bwd_dx
 *           does the first and the last in one kernel).
 * In the bf16-storage mode (ksr_set_precision) the [N, C] tensors of ksr_chan_dw_fwd / _bwd and ksr_chan_apply are bf16 like
 * those of every other kernel of the chain (channel counts, row lengths and offsets then multiples of 8); the per-image
 * matrices stay fp32.
 * ------------------------------------------------------------------------------------------- */

/* Depthwise 3x3 (zero pad 1) over `groups` (1..3) blocks of 64 channels of an NHWC tensor with ld channels
 * per pixel: q_dw / k_dw / v_dw (conv_sep_kan_attention.py:16-37).  dw_w [groups*64][9], dw_b [groups*64]. */
int ksr_chan_dw_fwd(const float* in, const float* dw_w, const float* dw_b, float* out, int ld, int groups, int B,
                    int H, int W, ksr_stream_t stream);
/* Its data gradient (transposed depthwise conv). */
int ksr_chan_dw_bwd(const float* dout, const float* dw_w, float* din, int ld, int groups, int B, int H, int W,
                    ksr_stream_t stream);
/* The same for q | k | v at once (ld = 192, groups = 3) with the data gradient of the three 1x1 projections
 * (conv_sep_kan_attention.py:16-37, q_proj / k_proj / v_proj) in the kernel's epilogue: din [N,192] = convT(dout),
 * dx [N,64] = din W, pw_w = W [192][64] (the projections' weights, q | k | v).  Replaces ksr_chan_dw_bwd followed by
 * ksr_pointwise_bwd(n_out = 192) in the bf16-storage mode (the only mode it supports), bit-equal to that pair. */
int ksr_chan_dw_bwd_dx(const float* dout, const float* dw_w, const float* pw_w, float* din, float* dx, int B, int H, int W,
                       ksr_stream_t stream);

/* Batched Gram product over pixels (the q k^T of conv_sep_kan_attention.py:65 and the dy^T v of its
 * backward): for each of nb entries (consecutive runs of tokens_per_entry pixels)
 *   out[e][j][k] = sum_n dz[n][dz_c0+j] * a[n][a_c0+k],   tail[e][0:J] = sum_n dz[n][j],
 * and with want_sq (J % 32 == 0): tail[e][J:2J] = sum_n dz[n][j]^2, tail[e][2J:2J+K] = sum_n a[n][k]^2
 * (the squared norms F.normalize needs, :63-64).  J % 4 == 0, K 32 or 64.  tail: nb * (want_sq ? 2J+K : J)
 * floats, may be NULL without want_sq.  slab: ksr_gram_slab_floats(...) floats. */
int64_t ksr_gram_slab_floats(int J, int K, int nb, int64_t tokens_per_entry);
int ksr_gram_batched(const float* dz, int dz_ld, int dz_c0, int J, const float* a, int a_ld, int a_c0, int K,
                     int64_t tokens_per_entry, int nb, float* slab, float* out, float* tail, int want_sq,
                     ksr_stream_t stream);

/* Per image: from G = q^T k [B][64][64] and tail (sum q | sum q^2 | sum k^2, 192 floats per image):
 * S = G / (max(|q|,1e-12) max(|k|,1e-12)), A = softmax over the last axis (:63-65), nrm [B][128] = raw
 * norms, M [B][n_out][64] = a_proj.weight * A (:67-69 folded). */
int ksr_chan_soft_fwd(const float* G, const float* tail, const float* a_proj_w, float* A, float* S, float* nrm,
                      float* M, int n_out, int B, ksr_stream_t stream);
/* Backward of the per-image algebra from Gd [B][n_out][64] = dy^T v: Mqk [B][128][128] such that
 * [dq | dk][n] = Mqk [q | k][n], and dWa [n_out][64] = gradient of a_proj.weight (dWa_part: B*n_out*64
 * floats of workspace). */
int ksr_chan_soft_bwd(const float* Gd, const float* A, const float* S, const float* nrm, const float* a_proj_w,
                      float* Mqk, float* dWa_part, float* dWa, int n_out, int B, ksr_stream_t stream);
/* out[i] = sum_b in[b*stride + i], i < n, in a fixed order (bias gradient from per-image column sums). */
int ksr_sum_entries(const float* in, int B, int64_t stride, int n, float* out, ksr_stream_t stream);

/* Per-image 1x1 conv: out[n, out_c0 : out_c0+n_out] = Wm[b(n)] * in[n, in_c0 : in_c0+n_in] (+ bias [n_out]),
 * Wm [B][n_out][n_in] row-major, or [B][n_in][n_out] when transposed != 0.  Supported n_in -> n_out:
 * 64 -> {40, 64, 160, 336};  {40, 64, 160, 336} -> 64 transposed;  128 -> 128. */
int ksr_chan_apply(const float* in, int in_ld, int in_c0, int n_in, const float* Wm, int transposed,
                   const float* bias, float* out, int out_ld, int out_c0, int n_out, int B, int HW,
                   ksr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training step support (DefaultPipeline.training_step / configure_optimizers,
 * src/ml/pipelines/default.py:59-84, and the implicit DDP gradient all-reduce, src/cli/train.py:58-62).
 */

/* loss = mean |pred[:,0] - tgt[:,0]| (nn.L1Loss on the Y plane, default.py:79) and its gradient
 * grad[b, :] = sign(pred - tgt) * grad_scale / numel.  slab256: 256 floats; loss_out: 1 float (device). */
int ksr_l1_y_loss(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride,
                  float* grad, int64_t grad_batch_stride, int B, int64_t plane_elems, float grad_scale,
                  float* slab256, float* loss_out, ksr_stream_t stream);

/* ksr_l1_y_loss for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments without the workspace: the 256
 * partial sums live in a 1 KB array inside the library's device image, one copy per GPU, held until the process ends; its
 * address is looked up (hipGetSymbolAddress) on the first call on each device - make that first call outside stream
 * capture, with the device of `stream` current.  Calls no hipMalloc / hipFree.  B * plane_elems <=
 * KSR_L1_STATIC_MAX_ELEMS (2^32), larger inputs are rejected with KSR_ERR_ARG.  All calls on one GPU share the array:
 * calls that overlap on one GPU (two streams, two threads) race. */
#define KSR_L1_STATIC_MAX_ELEMS ((int64_t)1 << 32)
int ksr_l1_y_loss_static(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride,
                         float* grad, int64_t grad_batch_stride, int B, int64_t plane_elems, float grad_scale,
                         float* loss_out, ksr_stream_t stream);

/* torch.optim.Adam step (default.py:61) over flat buffers; grads are multiplied by grad_scale first
 * (1/world_size after a sum all-reduce).  step counts from 1. */
int ksr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  ksr_stream_t stream);

/* The same step for recorded HIP graphs, whose by-value arguments are frozen: the step count (from 1) and the learning
 * rate are read from device memory at execution time.  Bias corrections are evaluated on the device in fp64. */
int ksr_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      const uint64_t* step_dev, float grad_scale, ksr_stream_t stream);

/* Dropout stream id from device memory, for the same reason: while the CALLING THREAD has a device word set here, every
 * entry point with a (p_drop, seed) pair draws its masks from seed + word * 0x9E3779B97F4A7C15 (the word is read when
 * the kernel executes); the forward and backward entry points of one step must see the same word.  NULL (the default)
 * switches this off.  The train-mode dropout of gazetr.py:37-49,77 has no such notion: torch's generator advances. */
int ksr_set_dropout_seed_source(const uint64_t* device_word);

/* Gradient exchange: one RCCL communicator per process (one process per GPU).  Rank 0 creates the
 * 128-byte id, every rank calls ksr_comm_init with it; ksr_allreduce_sum is an in-place fp32 sum. */
int ksr_comm_unique_id(void* out128);
int ksr_comm_init(int rank, int nranks, const void* unique_id128);
int ksr_allreduce_sum(float* buf, int64_t n, ksr_stream_t stream);
int ksr_comm_nranks(void);
int ksr_comm_destroy(void);

/* ---------------------------------------------------------------------------------------------
 * UNet hyper-encoder, forward (conv_sep_kan/unet.py:7-42; blocks conv_sep_kan/unet_blocks.py:9-178).
 * Activations are NHWC fp32 [B,H,W,C].
 */

/* Direct convolution (nn.Conv2d of DoubleConv unet_blocks.py:17-27, the stride-2 reflect-padded projections of
 * SelfAttention :100-127, the 1x1 convs :117,129 and OutConv :85).  Input = channel concatenation of src0 [B,Hin,Win,c0]
 * and src1 [B,Hin,Win,c1] (torch.cat of Up.forward :77; src1 NULL with c1 = 0); with in_scale/in_shift, src0 is read
 * as relu(x*scale[c]+shift[c]) (the BatchNorm + ReLU of the producing conv, :21-22).  weight [ks*ks][c0+c1][cout]
 * (tap-major, cout innermost); w_image_stride != 0 selects per-image weights.  ks 1 or 3 (padding ks/2, zero or
 * reflect), stride 1 or 2, cout % 4 == 0.  stat_partials (or NULL): ksr_unet_conv_partials(B,Hout,Wout,c0+c1,cout) rows of
 * [2][cout] floats, the per-workgroup sum and sum of squares of the output for BatchNorm's batch statistics. */
int64_t ksr_unet_conv_partials(int B, int Hout, int Wout, int cin, int cout);
int ksr_unet_conv_fwd(const float* src0, int c0, const float* src1, int c1, const float* in_scale,
                      const float* in_shift, const float* weight, int64_t w_image_stride, const float* bias,
                      float* out, int cout, float* stat_partials, int ks, int stride, int reflect, int B, int Hin,
                      int Win, ksr_stream_t stream);

/* nn.BatchNorm2d (unet_blocks.py:21,27) reduced to y = x*scale + shift.  train != 0: batch mean / biased variance from
 * the n_partials rows written by ksr_unet_conv_fwd over `count` values per channel, and the running statistics move
 * by `momentum` (unbiased variance), as torch does; train == 0: the running statistics are used and left alone.
 * saved_mean_invstd (or NULL): [2][C], the mean and 1/sqrt(var+eps) that were used, for the backward. */
int ksr_unet_bn_finalize(const float* stat_partials, int n_partials, int64_t count, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         int train, float* scale, float* shift, float* saved_mean_invstd, int C,
                         ksr_stream_t stream);

/* out = relu(raw*scale + shift) (unet_blocks.py:27-28) and, with pooled != NULL, pooled [B,H/2,W/2,C] =
 * nn.MaxPool2d(2)(out) (:40).  C % 4 == 0. */
int ksr_unet_bn_relu_fwd(const float* raw, const float* scale, const float* shift, float* out, float* pooled,
                         int B, int H, int W, int C, ksr_stream_t stream);

/* nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2) (unet_blocks.py:59-62): in [B,h,w,cin] ->
 * out [B,2h,2w,cout]; weight [dy*2+dx][cin][cout], bias [cout]. */
int ksr_unet_convt_fwd(const float* in, const float* weight, const float* bias, float* out, int cin, int cout,
                       int B, int h, int w, ksr_stream_t stream);

/* SelfAttention's per-image algebra (unet_blocks.py:141-177).  gram [B][zc][zc] = z^T z per image for z [B,n,zc] =
 * q (c channels) | k (c) | a (c/2) at half resolution (computed by ksr_unet_wgrad: ks = 1, a = g = z, per_image).
 * Forms the L2-normalised products q a^T and a k^T (:154-156,161,164), their softmaxes scaled by the per-head
 * temperatures, attn_a attn_k (:167-169) and folds v_proj (times vscale: the reference multiplies v by the image
 * width, :132,137-138) and project_out (:177) in: T [B][c][c] ([in][out]) with attention(x)[b,p,:] = x[b,p,:] T[b],
 * applied by ksr_unet_conv_fwd (ks = 1, w_image_stride = c*c).  c in {4, 8, 16, 32}. */
int ksr_unet_att_mix(const float* gram, int zc, int c, int heads, int B, const float* temp_a, const float* temp_v,
                     const float* wv, const float* wp, float vscale, float* T, ksr_stream_t stream);

/* ---- UNet backward.  Data gradients of the stride-1 / 1x1 convs and of the transposed conv are ksr_unet_conv_fwd
 * calls on re-laid-out weights (3x3: taps reversed, in/out swapped; ConvTranspose2d: ks = 2, stride = 2). */

/* Weight gradient, reduced over pixels and (unless per_image) images, in the forward weight layout:
 * mode 0: out[tap][ci][co] = sum a[b, map(p, tap)][ci] g[b,p][co] for the conv ks/stride/reflect of
 *         ksr_unet_conv_fwd (a = a0 | a1 concatenated, a0 optionally read as relu(x*a_scale+a_shift));
 * mode 1: ConvTranspose2d 2x2: out[dy*2+dx][ci][co] = sum a[b,y,x][ci] g[b,2y+dy,2x+dx][co];
 * mode 2: out[co] = sum g[b,p][co] (bias gradients).
 * a [B,Ha,Wa,ca0(+ca1)], g [B,Hg,Wg,cg]; slab: ksr_unet_wgrad_slab_floats(...) floats; per_image: out gets a
 * leading [B] axis (the per-image matrices of the attention backward).  add_ones (modes 0, 1): a channel of ones is
 * appended to a, i.e. out is [tap][ca+1][cg] and row ca of every tap holds sum g (the bias gradient of that tap's
 * pixels) - pass ca+1 as `ca` to ksr_unet_wgrad_slab_floats. */
int64_t ksr_unet_wgrad_slab_floats(int mode, int ks, int ca, int cg, int B, int Ha, int Wa, int Hg, int Wg);
int ksr_unet_wgrad(const float* a0, int ca0, const float* a1, int ca1, const float* a_scale, const float* a_shift,
                   const float* g, int cg, float* slab, float* out, int per_image, int add_ones, int mode, int ks,
                   int stride, int reflect, int B, int Ha, int Wa, int Hg, int Wg, ksr_stream_t stream);

/* Backward of relu(BatchNorm2d(raw)) (unet_blocks.py:21-22,27-28): dout = gradient of the activated output;
 * writes dgamma, dbeta [C] and dx = gradient of raw.  train != 0 differentiates through the batch statistics.
 * scale/shift/saved_mean_invstd as produced by ksr_unet_bn_finalize; partials: ksr_unet_bn_bwd_partials(npix)
 * rows of [2][C]; coef: 3*C floats of workspace. */
int64_t ksr_unet_bn_bwd_partials(int64_t npix);
int ksr_unet_bn_relu_bwd(const float* dout, const float* raw, const float* scale, const float* shift,
                         const float* gamma, const float* saved_mean_invstd, int train, float* partials,
                         float* coef, float* dgamma, float* dbeta, float* dx, int64_t npix, int C,
                         ksr_stream_t stream);

/* KAN cell (ConvSepKanCell.forward, conv_sep_kan/conv_sep_kan_cells.py:136-156, I = 1) whose per-pixel parameter vector
 * is the output of a 1x1 conv 4 -> P (OutConv of the UNet hyper-encoder, conv_sep_kan/unet.py:28,41), w[n,:] = lb + lw y[n,:],
 * computed on the fly instead of read: x [B,1,H,W], y [B*H*W,4], lw [P][4] (= the conv weight [P,4,1,1]), lb [P],
 * P = O (n_coef + 2) <= 512; out as ksr_kan_cell_fwd.  The backward writes dy [B*H*W,4] and dlwb [5][P] (rows 0..3: the
 * gradient of lw transposed, row 4: of lb); slab: ksr_kan_lin_slab_floats(...) floats of workspace. */
int64_t ksr_kan_lin_slab_floats(int B, int H, int W, int O, int n_coef);
int ksr_kan_lin_fwd(const float* x, const float* y, const float* lw, const float* lb, float* out, int B, int O, int H,
                    int W, int n_coef, const float* knots, int n_knots, int spline_order, int shuffle_r,
                    int64_t out_batch_stride, ksr_stream_t stream);
int ksr_kan_lin_bwd(const float* x, const float* y, const float* lw, const float* lb, const float* gout, float* dy,
                    float* slab, float* dlwb, int B, int O, int H, int W, int n_coef, const float* knots, int n_knots,
                    int spline_order, int shuffle_r, int64_t gout_batch_stride, ksr_stream_t stream);

/* One launch for a table of small strided re-layouts (no counterpart call in the reference: there they are the
 * implicit layouts of nn.Conv2d / nn.ConvTranspose2d weights, conv_sep_kan/unet_blocks.py:9-28,46-63,91-129, and
 * autograd's chain rule through them).  Entry e, over the 4-D index space shape (row-major, `count` elements):
 *   dst[sum_d i_d dstride_d] = sum_{r<nred} src[sum_d i_d sstride_d + r rstride] * (src2 ? src2[sum_d i_d s2stride_d + r r2stride] : 1)
 * Strides are in floats and may be negative (flips) or zero.  `descs` and `blocks` live in device memory; blocks[k]
 * names the entry and the first of the 256 consecutive elements workgroup k handles (the caller lists
 * ceil(count/256) blocks per entry).  Entries of one table must not read what another entry of it writes. */
typedef struct KsrGatherDesc {
    const float* src;
    const float* src2;
    float* dst;
    int32_t shape[4];
    int32_t sstride[4];
    int32_t s2stride[4];
    int32_t dstride[4];
    int32_t nred, rstride, r2stride, count;
} KsrGatherDesc;
typedef struct KsrGatherBlock { int32_t desc, first; } KsrGatherBlock;
int ksr_gather_batch(const KsrGatherDesc* descs, const KsrGatherBlock* blocks, int nblocks, ksr_stream_t stream);

/* Sum of the gradients that meet at an encoder level: out = g0 + g1 (optional) + MaxPool2d(2) backward of
 * g_pooled (optional; routed to the first maximum of each window of act, the activated output). */
int ksr_unet_grad_merge(const float* g0, const float* g1, const float* g_pooled, const float* act, float* out,
                        int B, int H, int W, int C, ksr_stream_t stream);

/* Data gradient of the 3x3 stride-2 reflect-padded projections (unet_blocks.py:100-127): dz [B,Ho,Wo,cz] ->
 * dx [B,H,W,cin]; weight [9][cz][cin]. */
int ksr_unet_s2_dgrad(const float* dz, int cz, const float* weight, float* dx, int cin, int B, int H, int W,
                      ksr_stream_t stream);

/* Backward of ksr_unet_att_mix from dT [B][c][c]: D [B][zc][zc] with dz[b,n,:] = z[b,n,:] D[b] (applied by
 * ksr_unet_conv_fwd, ks = 1) and dparams [B][2c^2 + 2 heads] = per-image gradients of v_proj | project_out |
 * temperature_a | temperature_v (to be summed over images). */
int ksr_unet_att_mix_bwd(const float* gram, const float* dT, int c, int heads, int zc, int B,
                         const float* temp_a, const float* temp_v, const float* wv, const float* wp, float vscale,
                         float* D, float* dparams, ksr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Blocks of the SeemoRe backbone (conv_sep_kan/seemore.py), forward only, NHWC fp32.  The reference's top class
 * SeemoRe (:13-77) and its RME / ResGroup (:84-107, :162-181) raise in their constructors (RME passes mlp_ratio= to a
 * GatedFFN that has no such argument, :176 vs :355-358), so the drop-in surface is the constructible blocks:
 * StripedConv2d :325-341, GatedFFN :354-382, LayerNorm(channels_first) :386-410, StripedConvFormer :134-155, SME :114-129,
 * Expert :289-302, Router :305-318, MoELayer :261-285, MoEBlock :188-258 (host side: kan_sr/seemore.py).
 * Tensors are [B, H, W, ld] with the operand's channels starting at c0 (torch.chunk halves are read in place).
 * ------------------------------------------------------------------------------------------- */

/* nn.Conv2d (dense or depthwise, any kernel shape / stride / zero padding) with optional exact GELU (nn.GELU()) after it,
 * optional elementwise factor `mul` on the input (q * v of seemore.py:154, x * gate of :379) and optional residual added
 * after the activation (seemore.py:127-128).  w: kernel layout, dense [KH*KW][Cin][Cout] / depthwise [KH*KW][Cout].
 * Cout % 8 == 0, Cin % 4 == 0, row lengths and first channels multiples of 4. */
int ksr_sm_conv_fwd(const float* in, int in_ld, int in_c0, const float* mul, int mul_ld, int mul_c0, const float* w,
                    const float* bias, const float* res, int res_ld, int res_c0, float* out, int out_ld, int out_c0,
                    int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                    int depthwise, int act, ksr_stream_t stream);
/* LayerNorm(data_format="channels_first").forward (seemore.py:402-410) over 64 channels: in / out [npix, 64]. */
int ksr_sm_layernorm_fwd(const float* in, const float* gain, const float* shift, float* out, int64_t npix, float eps,
                         ksr_stream_t stream);
/* MoEBlock.calibrate's tail (seemore.py:243-244): out [B,H,W,64] = res + F.interpolate(src [B,hs,ws,64], size=(H,W),
 * mode="bilinear", align_corners=False). */
int ksr_sm_resize_add_fwd(const float* src, int hs, int ws, const float* res, int res_ld, int res_c0, float* out, int B,
                          int H, int W, ksr_stream_t stream);
/* Router.forward + the softmax / top-k of MoELayer.forward (seemore.py:317-318, :270-277): weights [B, E] = softmax over the
 * experts of router_w [E,64] applied to the per-image channel means, all but the topk largest set to 0; logits_out [B, E]
 * (optional) = Router.forward's output.  E <= 8. */
int ksr_sm_route_fwd(const float* in, int in_ld, int in_c0, const float* router_w, int E, int topk, int B, int64_t HW,
                     float* weights, float* logits_out, ksr_stream_t stream);
/* The sum of MoELayer.forward (seemore.py:273-285) with Expert.forward (:298-302) inside: out [B*HW, 64] = x + sum_e
 * weights[b, e] * conv_3_e(conv_2_e(k) * conv_1_e(x)).  ptrs: HOST array of 6 device pointers per expert (conv_1.weight
 * [low,64], conv_1.bias, conv_2.weight [low,64], conv_2.bias, conv_3.weight [64,low], conv_3.bias); low: HOST array. */
int ksr_sm_experts_fwd(const float* x, int x_ld, int x_c0, const float* k, int k_ld, int k_c0, const float* weights,
                       const float* const* ptrs, const int* low, int n_experts, float* out, int B, int64_t HW,
                       ksr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KSR_H */
