/* vad_hip.h — C ABI of libvad_hip.so: the MI355X (gfx950) per-frame anomaly-scoring hot path.
 *
 * The reference (KuldeepChoksi/video-anomaly-detection) has no FFI: its boundary is the Python
 * nn.Module API of models/.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference root).  Everything is plain pointers and sizes; `stream` is a
 * hipStream_t passed as void*.  All device pointers are fp32 unless stated.  Calls are
 * asynchronous on `stream`, never allocate and never synchronise (except vad_prof_read).
 *
 * Return value: VAD_OK or a negative VAD_ERR_*; vad_last_error() gives the text.
 */
#ifndef VAD_HIP_H
#define VAD_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VAD_ABI_VERSION 3   /* 2: the arithmetic mode is an argument of every packer / launcher; no process-wide switch.
                             * 3: packed-blob layout of round 3 (tail slot carries the GEMM form the fused dec4 kernel reads, dec4.0 is fp32 in
                             *    every blob, latent / hidden widths zero-padded to the channel tiling): blobs and libraries of ABI 2 are
                             *    rejected by vad_blob_precision and by the device-side tag check */
#define VAD_OK 0
#define VAD_ERR_ARG (-1)   /* bad argument / unsupported shape */
#define VAD_ERR_HIP (-2)   /* HIP runtime error */
#define VAD_ERR_WS (-3)    /* workspace too small */

#define VAD_ACT_NONE 0
#define VAD_ACT_LEAKY 1    /* LeakyReLU(0.2): models/autoencoder.py:41, models/video_autoencoder.py:195 */
#define VAD_ACT_RELU 2     /* ReLU: models/autoencoder.py:106 */

/* Arithmetic of the 3x3 / transposed convolutions.  It is an ARGUMENT (`precision`) of every entry point whose operand
 * layout or kernel depends on it - never process state - so two models with different modes can be scored from two
 * threads at once (the reference's UI calls one global model from worker threads, main.py:50,274).  Weights must be packed
 * and launched with the same value; the model blobs carry it in their header (see vad_img_pack).
 *   VAD_PREC_FP32  exact fp32: v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain (the parity path);
 *   VAD_PREC_SPLIT split fp16: every fp32 operand v is used as hi = fp16(v), lo = fp16((v-hi)*2^11) and a*b is evaluated
 *      as ah*bh + (ah*bl + al*bh)*2^-11 with three v_mfma_f32_32x32x16_f16 and fp32 accumulation.  fp16 products are
 *      exact in fp32, so each product carries 22 significant bits (fp32 has 24); activations and outputs in HBM stay
 *      fp32.  Needs |activation| < 65504.  Opt-in; gated by the same golden-vector and trained-model tests. */
#define VAD_PREC_FP32 0
#define VAD_PREC_SPLIT 1
/*   VAD_PREC_BF16  bf16 operands (round to nearest even), one v_mfma_f32_32x32x16_bf16 per 16 channels, fp32 accumulate -
 *      TRAINING ONLY (BASELINE.json configs[4] names bf16): accepted by vad_conv3x3, vad_convt2x2, vad_train_pack_* and the
 *      vad_conv_wgrad and the vad_*_train_fwd_bwd entry points; master weights, BatchNorm, statistics, loss, the first /
 *      last layer and Adam stay fp32 (weight gradients: bf16 operands, fp32 accumulation).
 *      8 significant bits: the scoring entry points and host packers reject it (scores would miss the 1e-4 bar by 60x). */
#define VAD_PREC_BF16 2
/*   VAD_PREC_BF16S bf16 STORAGE as well (training only, the mode `VideoTrainer(precision="bf16")` runs): every activation and
 *      activation-gradient tensor between the kernels - conv outputs before BatchNorm, pooled activations, ConvLSTM operand
 *      and gate buffers, gradient scratch - is bf16 in HBM (half the bytes of the HBM-bound BatchNorm passes, MFMA operands
 *      staged without conversion); arithmetic inside every kernel, BatchNorm statistics (from the fp32 accumulators), cell
 *      states, parameters, parameter gradients, loss and Adam stay fp32.  Layer entry points that take `precision` read /
 *      write bf16 tensors through their `float*` arguments in this mode (element strides stay in elements). */
#define VAD_PREC_BF16S 3
/*   VAD_PREC_WINO  Winograd F(2x2,3x3), scoring only, OPT-IN (`model.precision = "winograd"`): every 3x3 convolution behind
 *                  the first layer computes a 2x2 block of outputs from 16 instead of 36 products per input channel
 *                  (csrc/conv_wino.hip) on the exact-fp32 matrix pipe - all-fp32 arithmetic, but another rounding order than the
 *                  direct form, so NOT bit-identical to VAD_PREC_FP32 (scores differ by ~1e-6 relative; every parity gate
 *                  holds at 1e-5).  Everything else (first layer, transposed convolutions, tails, ConvLSTM cell) is the
 *                  VAD_PREC_FP32 arithmetic.  bench.py's `value` is always VAD_PREC_FP32. */
#define VAD_PREC_WINO 4

int vad_abi_version(void);
const char* vad_last_error(void);   /* per thread */

/* ------------------------------------------------------------------ weight packing (host, CPU)
 * Folds eval-mode BatchNorm2d (eps 1e-5; y = (x-mean)/sqrt(var+eps)*gamma+beta) into the preceding
 * conv in fp64, rounds once to fp32 and re-orders for the kernels' MFMA B-operand loads.
 * bn == NULL means "no BatchNorm after this conv"; bn = {gamma, beta, running_mean, running_var}. */

/* Conv2d k3 p1 weight OIHW (Cout,Cin,3,3) -> [9][ceil(Cin/8)][Cout][8]; bias_out[Cout].
 * Replaces nn.Conv2d+nn.BatchNorm2d pairs of models/autoencoder.py:38-79,103-139 and
 * models/video_autoencoder.py:46-52,191-215. */
size_t vad_pack_conv3x3_floats(int cout, int cin);
int vad_pack_conv3x3(const float* w_oihw, const float* bias, const float* const* bn,
                     int cout, int cin, int precision, float* w_packed, float* bias_out);

/* First-layer form (Cin == 3, K = 27 padded to 28): -> [14][2][Cout], followed by the split-fp16 form (both modes). */
size_t vad_pack_conv3x3_c3_floats(int cout);
int vad_pack_conv3x3_c3(const float* w_oihw, const float* bias, const float* const* bn,
                        int cout, float* w_packed, float* bias_out);

/* ConvTranspose2d k2 s2 weight IOHW (Cin,Cout,2,2) -> [4][Cin/8][Cout][8] (quadrant q = 2*a+b).
 * Replaces models/autoencoder.py:104,113,122,131 and models/video_autoencoder.py:244-260. */
size_t vad_pack_convt2x2_floats(int cin, int cout);
int vad_pack_convt2x2(const float* w_iohw, const float* bias, const float* const* bn,
                      int cin, int cout, int precision, float* w_packed, float* bias_out);

/* Conv2d k1 (VideoAutoencoder.proj, models/video_autoencoder.py:311) -> [Cin/8][Cout][8]. */
size_t vad_pack_conv1x1_floats(int cout, int cin);
int vad_pack_conv1x1(const float* w_oihw, const float* bias, int cout, int cin,
                     float* w_packed, float* bias_out);

/* ------------------------------------------------------------------ layer kernels (device)
 * Activations are NHWC fp32.  H, W are the INPUT spatial size.  *_fs = frame stride in floats
 * (0 means dense: H*W*C of that tensor). */

/* x NCHW [N,3,H,W] -> conv3x3(3->Cout)+bias+act(+maxpool2) -> NHWC.  Cout multiple of 32. */
int vad_conv3x3_c3(const float* x_nchw, const float* w_packed, const float* bias, float* out_nhwc,
                   int n, int h, int w, int cout, int act, int pool, void* stream);

/* Fused Encoder.enc1 (models/autoencoder.py:38-46): conv3x3(3->32)+BN+LeakyReLU, conv3x3(32->32)+BN+LeakyReLU,
 * MaxPool2d(2,2) in ONE launch; the 32-channel full-resolution map stays in LDS.  w0/b0 from
 * vad_pack_conv3x3_c3 (cout 32), w1/b1 from vad_pack_conv3x3 (32,32).  out NHWC [N,H/2,W/2,32]. */
int vad_conv3x3_c3_fused(const float* x_nchw, const float* w0, const float* b0, const float* w1, const float* b1,
                         float* out_nhwc, int n, int h, int w, int precision, void* stream);

/* NHWC conv3x3 p1 + bias + act (+ MaxPool2d(2,2) when pool != 0).  Cin multiple of 8 (32 for
 * speed), Cout multiple of 32.  pool needs even H, W. */
int vad_conv3x3(const float* in_nhwc, long long in_fs, const float* w_packed, const float* bias,
                float* out_nhwc, long long out_fs, int n, int h, int w, int cin, int cout,
                int act, int pool, int precision, void* stream);

/* Winograd F(2x2,3x3) form of vad_conv3x3 on the exact-fp32 matrix pipe (csrc/conv_wino.hip): OPT-IN arithmetic - all fp32,
 * 16 instead of 36 multiplies per 2x2 outputs and input channel, NOT bit-identical to vad_conv3x3 (fp32 rounding differs).
 * Same operation as the nn.Conv2d(k3,p1)+BatchNorm2d+activation(+MaxPool2d) pairs of models/autoencoder.py:49-79,103-128.
 * w_packed from vad_pack_conv3x3_wino ([16][Cin/8][Cout][8], U = G g G^T).  Cin, Cout multiples of 32; even H, W. */
size_t vad_pack_conv3x3_wino_floats(int cout, int cin);
int vad_pack_conv3x3_wino(const float* w_oihw, const float* bias, const float* const* bn, int cout, int cin,
                          float* w_packed, float* bias_out);
int vad_conv3x3_wino(const float* in_nhwc, long long in_fs, const float* w_packed, const float* bias,
                     float* out_nhwc, long long out_fs, int n, int h, int w, int cin, int cout,
                     int act, int pool, void* stream);
/* One ConvLSTMCell step (models/video_autoencoder.py:54-85) with the gate convolution in Winograd form: arguments as
 * vad_convlstm_step, w_packed = vad_pack_conv3x3_wino of the (4*hid, cin_x+hid, 3, 3) weight, cin_x == hid, plus
 * z_ws = n*h*w*4*hid floats of scratch for the gate pre-activations (the cell is a second, pointwise launch). */
int vad_convlstm_step_wino(const float* x, long long x_fs, const float* h_prev, long long h_prev_fs, const float* c_prev,
                           const float* w_packed, const float* bias, float* h_out, long long h_out_fs, float* c_out,
                           float* z_ws, int n, int h, int w, int cin_x, int hid, void* stream);

/* NHWC ConvTranspose2d k2 s2 + bias + act: [N,H,W,Cin] -> [N,2H,2W,Cout]. */
int vad_convt2x2(const float* in_nhwc, long long in_fs, const float* w_packed, const float* bias,
                 float* out_nhwc, long long out_fs, int n, int h, int w, int cin, int cout,
                 int act, int precision, void* stream);

/* NHWC 1x1 conv + bias (no activation). */
int vad_conv1x1(const float* in_nhwc, const float* w_packed, const float* bias, float* out_nhwc,
                long long npix, int cin, int cout, void* stream);

/* One ConvLSTMCell step (models/video_autoencoder.py:54-85), gates fused in the conv epilogue.
 * w_packed = vad_pack_conv3x3 of the (4*hid, cin_x+hid, 3, 3) weight.  x [N,H,W,cin_x] (frame
 * stride x_fs), h_prev [N,H,W,hid] (frame stride h_prev_fs), c_prev dense (both NULL mean zeros,
 * models/video_autoencoder.py:87-91).  Writes h_out (frame stride h_out_fs) and c_out (dense;
 * may alias c_prev). */
int vad_convlstm_step(const float* x, long long x_fs, const float* h_prev, long long h_prev_fs,
                      const float* c_prev, const float* w_packed, const float* bias, float* h_out, long long h_out_fs,
                      float* c_out, int n, int h, int w, int cin_x, int hid, int precision, void* stream);

/* Scoring tails.  x is the ORIGINAL input, NCHW [N,3,H2,W2] with H2 = output size.
 * partials: [N][vad_score_partials(H2,W2)] per-tile sums of sum_c (x-recon)^2.
 * recon_nchw (NCHW [N,3,H2,W2]) and errmap ([N,H2,W2], channel mean) may be NULL.
 * Replaces models/autoencoder.py:211-221 and models/video_autoencoder.py:368-384. */
int vad_score_partials(int kind /*0: conv3x3 tail, 1: convT tail*/, int h2, int w2);
/* Conv2d(cin->3) weight OIHW (3,cin,3,3) -> [cin/4][9 taps][4 channels][3 outputs] (per-lane weight rows), followed for
 * cin == 32 by the GEMM form the fused kernel below reads (1024 floats at offset (cin/4)*108: [m = tap*3+co, 32 rows][lane
 * half][16 channel steps]). */
size_t vad_pack_conv3x3_to3_floats(int cin);
int vad_pack_conv3x3_to3(const float* w_oihw, int cin, float* w_packed);
/* Conv2d(32->3) k3 p1 + Tanh (models/autoencoder.py:134-135) on NHWC [N,H2,W2,32]. */
int vad_conv3x3_to3_score(const float* in_nhwc, const float* w_packed /*vad_pack_conv3x3_to3*/,
                          const float* bias3, const float* x_nchw, float* partials,
                          float* recon_nchw, float* errmap, int n, int h2, int w2, int cin,
                          void* stream);
/* dec4 block + scoring in ONE launch: ConvTranspose2d(32->32, k2 s2) + BatchNorm + ReLU, Conv2d(32->3, k3 p1) + Tanh,
 * squared error, channel mean, per-row partial sums (models/autoencoder.py:131-139, 211-221).  in_nhwc [N,H,W,32] is the
 * dec3.3 output; the frame is 2H x 2W; the 32-channel full-resolution map between the two layers is never stored.
 * wt_packed / bt: vad_pack_convt2x2(..., VAD_PREC_FP32) of dec4.0 with dec4.1 folded; w2_gemm: vad_pack_conv3x3_to3's
 * second form; partials: [N][vad_dec4_score_partials(2H, 2W)].  Exact fp32 only. */
int vad_dec4_score_partials(int h2, int w2);
int vad_dec4_score(const float* in_nhwc, const float* wt_packed, const float* bt, const float* w2_gemm,
                   const float* bias3, const float* x_nchw, float* partials, float* recon_nchw, float* errmap,
                   int n, int h, int w, void* stream);
/* ConvTranspose2d(32->3) k2 s2 + Tanh (models/video_autoencoder.py:259-260) on NHWC [N,H,W,32]. */
/* t, clip_stride: activation frame n is scored against x frame (n / t) * clip_stride + n % t (sliding windows
 * over one video); t == 0 or t == clip_stride means frame n (independent clips). */
int vad_convt2x2_to3_score(const float* in_nhwc, const float* w_iohw /*[cin][3][2][2]*/,
                           const float* bias3, const float* x_nchw, float* partials,
                           float* recon_nchw, float* errmap, int n, int h, int w, int cin,
                           int t, int clip_stride, void* stream);
/* frame_scores[N] = sum(partials[n][:]) / (3*H2*W2) in a fixed order (bit-exact under any
 * batching); if seq_scores != NULL also seq_scores[N/t] = mean over t consecutive frames. */
int vad_score_finalize(const float* partials, int nparts, int n, int h2, int w2,
                       float* frame_scores, float* seq_scores, int t, void* stream);

int vad_nhwc_to_nchw(const float* in, float* out, int n, int h, int w, int c, void* stream);
int vad_nchw_to_nhwc(const float* in, float* out, int n, int h, int w, int c, void* stream);

/* ------------------------------------------------------------------ criteria
 * Replaces SSIMLoss.forward (utils/losses.py:51-93) and CombinedLoss.forward (utils/losses.py:114-121) for tensors
 * that need no gradient: pred/target are `planes` = B*C contiguous H x W fp32 planes (NCHW), zero padding
 * window_size/2, Gaussian sigma 1.5, C1 = 1e-4, C2 = 9e-4.  ONE pass over both inputs; no map is materialised.
 * workspace: vad_ssim_workspace_floats(planes,h,w) floats (0 = unsupported shape).
 * out3 (device): { 1 - mean(SSIM), mean((pred-target)^2), (1-alpha)*out3[1] + alpha*out3[0] }. */
size_t vad_ssim_workspace_floats(long long planes, int h, int w);
int vad_ssim_mse(const float* pred, const float* target, long long planes, int h, int w, int window_size,
                 float alpha, float* workspace, float* out3, void* stream);
/* Gradient of out3[2] = (1-alpha)*MSE + alpha*(1 - mean SSIM) with respect to pred (alpha = 1: SSIMLoss, alpha = 0: MSE),
 * times the upstream gradient grad_out (DEVICE scalar, so autograd needs no host synchronisation).  What autograd derives
 * for utils/losses.py:51-121 when train.py:41-46 back-propagates through the criterion.  Two fused passes; workspace:
 * vad_ssim_grad_workspace_floats floats (three adjoint maps). */
size_t vad_ssim_grad_workspace_floats(long long planes, int h, int w);
int vad_ssim_mse_backward(const float* pred, const float* target, long long planes, int h, int w, int window_size,
                          float alpha, const float* grad_out, float* workspace, float* grad_pred, void* stream);

/* ------------------------------------------------------------------ training-step kernels
 * The pieces of one optimisation step of train_video.py:44-65 (model.train(); MSELoss; backward; Adam), exact fp32,
 * NHWC activations.  What the reference gets from autograd is stated here as explicit forward/backward kernels.
 * All `ws` arguments are device scratch sized by the matching *_ws_floats function. */

/* Per-channel reductions over an NHWC tensor [npix][c] (c multiple of 4, <= 1024). */
size_t vad_chan_ws_floats(long long npix, int c);
/* Train-mode nn.BatchNorm2d statistics (torch defaults eps 1e-5, momentum 0.1): stats[0..c) = batch mean,
 * stats[c..2c) = 1/sqrt(biased var + eps); running_mean/var (nullable) updated with the unbiased variance. */
int vad_bn_stats(const float* y, long long npix, int c, float eps, float momentum, float* stats,
                 float* running_mean, float* running_var, float* ws, void* stream);
/* out[c] = sum over pixels (bias gradients). */
int vad_chan_sum(const float* g, long long npix, int c, float* out, float* ws, void* stream);
/* BatchNorm(batch stats) + activation (+ MaxPool2d(2,2)) of a conv output y [n,h,w,c]
 * (models/video_autoencoder.py:191-215,242-256 in train mode).  Destination element (frame, pixel, ch) is at
 * out + F*out_fs + pixel*out_ps + ch, with F = frame, or (frame % T)*B + frame / T when remap_t = T > 0 (time-major
 * ConvLSTM operand buffers).  out_fs / out_ps 0 = dense. */
int vad_bn_act_pool_fwd(const float* y, const float* stats, const float* gamma, const float* beta, float* out,
                        long long out_fs, int out_ps, int remap_t, int remap_b, int n, int h, int w, int c,
                        int act, int pool, void* stream);
/* Backward of the above: dout is addressed like `out`.  dy = gradient of the conv output, dense NHWC or (s2d != 0,
 * un-pooled layers) the space-to-depth view [n][h/2][w/2][4][c]; must not alias dout.  dgamma/dbeta [c]; ksums [2c]
 * scratch.  Two passes over y: per-channel sums of the routed gradient, then dy; the routed gradient is never stored. */
int vad_bn_act_pool_bwd(const float* y, const float* stats, const float* gamma, const float* beta, const float* dout,
                        long long dout_fs, int dout_ps, int remap_t, int remap_b, float* dy, int s2d,
                        float* dgamma, float* dbeta, float* ksums, float* ws, int n, int h, int w, int c, int act,
                        int pool, void* stream);
/* ConvLSTMCell gate math (models/video_autoencoder.py:73-83) on pre-activations z [nb*hw][4*hid] (order i,f,g,o): z is
 * overwritten with the activated gates (kept for the backward), c_out = f*c_prev + i*g, h = o*tanh(c_out) written to up
 * to two destinations (frame b, pixel q, channel j) -> h + b*fs + q*ps + j.  c_prev NULL = zeros. */
int vad_lstm_gates_fwd(float* z, const float* c_prev, float* c_out, float* h1, long long h1_fs, int h1_ps,
                       float* h2, long long h2_fs, int h2_ps, int nb, int hw, int hid, void* stream);
/* Backward through the same step: dh = dh1 + dh2 (nullable sources), dc_next nullable -> dz (pre-activation gradients)
 * and dc_prev (may alias dc_next). */
int vad_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c, const float* dh1, long long dh1_fs,
                       int dh1_ps, const float* dh2, long long dh2_fs, int dh2_ps, const float* dc_next, float* dz,
                       float* dc_prev, int nb, int hw, int hid, void* stream);
/* Weight gradient GEMM, K = pixels: a [n,h,w,cin], g [n,h,w,ncols].
 *   taps 9, layout 0: Conv2d k3 p1 weight gradient, dw OIHW (ncols, cin, 3, 3)
 *   taps 1, layout 1: ConvTranspose2d k2 s2 weight gradient from the space-to-depth output gradient
 *                     (ncols = 4*cout, column q*cout+co), dw IOHW (cin, cout, 2, 2)
 *   taps 1, layout 3: ConvTranspose2d(32->3) from the 32-column dpre of vad_convt_to3_mse, dw (32, 3, 2, 2)
 *   taps 1, layout 4: Conv2d k1 (VideoAutoencoder.proj) weight gradient, dw OIHW (ncols, cin, 1, 1) */
size_t vad_conv_wgrad_ws_floats(int n, int h, int taps, int cin, int ncols);
/* precision VAD_PREC_FP32: exact fp32 (v_mfma_f32_32x32x2_f32, two pixels per instruction); VAD_PREC_SPLIT (round 4): both
 * operands split into fp16 (hi, lo) pairs as they are packed, three v_mfma_f32_32x32x16_f16 per 16 pixels (22-bit products, fp32
 * accumulation; |a|, |g| < 65504 and g should sit in the fp16 range - the step scales its gradients, see vad_convt_to3_mse_t);
 * VAD_PREC_BF16: both operands rounded to bf16, v_mfma_f32_32x32x16_bf16 (16 pixels per instruction), fp32 accumulation and
 * fp32 split-K partials. */
int vad_conv_wgrad(const float* a, const float* g, float* dw, float* ws, int n, int h, int w, int cin, int ncols,
                   int taps, int layout, int precision, void* stream);
/* First-layer weight gradient: x NCHW [n,3,h,w], g [n,h,w,cout] -> dw OIHW (cout,3,3,3). */
size_t vad_conv_c3_wgrad_ws_floats(int n, int h, int cout);
int vad_conv_c3_wgrad(const float* x_nchw, const float* g, float* dw, float* ws, int n, int h, int w, int cout,
                      void* stream);
/* ConvTranspose2d(32->3,k2,s2)+Tanh+MSELoss(mean), forward and backward in one pass (models/video_autoencoder.py:259-260,
 * train_video.py:54-55): in [n,h,w,32], weight IOHW (32,3,2,2) as stored by torch, x NCHW [n,3,2h,2w].
 * loss[0] = mean((recon-x)^2); recon (NCHW), din [n,h,w,32], dpre32 [n*h*w][32] and dbias3 are optional outputs. */
size_t vad_convt_to3_mse_ws_floats(int n, int h, int w);
int vad_convt_to3_mse(const float* in_nhwc, const float* w_iohw, const float* bias3, const float* x_nchw, float* recon,
                      float* din, float* dpre32, float* loss, float* dbias3, float* ws, int n, int h, int w,
                      void* stream);
/* torch.optim.Adam step (train_video.py:175: weight_decay is L2 added to the gradient) over a flat buffer;
 * grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int vad_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, void* stream);
/* Device-side operand packing of the CURRENT parameters (no BatchNorm folding in train mode; the layout follows
 * `precision` like the host packers; VAD_PREC_WINO: both operands in Winograd form, vad_pack_conv3x3_wino_floats of room each):
 * fwd = the forward kernels' order (vad_pack_conv3x3 / vad_pack_convt2x2 / vad_pack_conv3x3_c3 layouts);
 * dgrad = the data-gradient operand: for conv3x3 a conv3x3 weight with cin/cout swapped and taps rotated (run it
 * through vad_conv3x3), for convT a 1x1 weight with K = 4*cout (run vad_conv1x1 on the space-to-depth gradient). */
int vad_train_pack_conv3x3(const float* w_oihw, int cout, int cin, float* fwd, float* dgrad, int precision, void* stream);
int vad_train_pack_convt2x2(const float* w_iohw, int cin, int cout, float* fwd, float* dgrad, int precision, void* stream);
int vad_train_pack_conv3x3_c3(const float* w_oihw, int cout, float* fwd, void* stream);
/* Conv2d(cin->3) k3 (ConvAutoencoder's last conv, models/autoencoder.py:134): fwd = vad_pack_conv3x3_to3 layout for the
 * scoring tail kernel; dgrad_c3 = the rotated weights in vad_pack_conv3x3_c3 form (vad_pack_conv3x3_c3_floats(cin) floats):
 * the data gradient is a 3->cin first-layer convolution of the pre-activation gradient. */
int vad_train_pack_conv3x3_to3(const float* w_oihw, int cin, float* fwd, float* dgrad_c3, void* stream);
/* Backward of Conv2d(32->3,k3,p1)+Tanh: recon = tanh(conv(in)) [n,3,h,w]; the upstream gradient is either drecon [n,3,h,w]
 * or (drecon NULL) that of nn.MSELoss against x, times grad_mul (a power of two; 1 = none, see vad_convt_to3_mse_t).  Outputs: dpre scratch [n,3,h,w], din [n,h,w,32], dw (3,32,3,3), db3 [3]. */
size_t vad_conv3x3_to3_bwd_ws_floats(int n, int h, int w, int cin);
int vad_conv3x3_to3_tanh_bwd(const float* in_nhwc, const float* recon, const float* x, const float* drecon,
                             const float* w_dgrad_c3, float* dpre, float* din, float* dw, float* db3, float* ws,
                             int n, int h, int w, int cin, float grad_mul, void* stream);
/* Conv2d k1 (cout, cin, 1, 1): fwd = vad_pack_conv1x1 layout; dgrad = the transposed 1x1 weight (K = cout, N = cin). */
int vad_train_pack_conv1x1(const float* w_oihw, int cout, int cin, float* fwd, float* dgrad, void* stream);

/* bf16-tensor forms of the kernels above (VAD_PREC_BF16S, `VideoTrainer(precision="bf16")`): io16 != 0 means every
 * activation / activation-gradient tensor argument (y, out, dout, dy, z, h1, h2, gates, dh1, dh2, dz, g, in, din, dpre32) is
 * bf16 in memory - strides stay in ELEMENTS - while statistics, cell states, parameters, their gradients and the loss stay
 * fp32; io16 == 0 is the fp32 entry point of the same name without `_t`.  The arithmetic inside is the same fp32 arithmetic
 * on the widened values, results rounded to bf16 (nearest even) where they are stored.  vad_conv3x3 / vad_convt2x2 /
 * vad_conv_wgrad / vad_train_pack_* take precision VAD_PREC_BF16S for the same tensors. */
int vad_chan_sum_t(const void* g, int io16, long long npix, int c, float* out, float* ws, void* stream);
int vad_bn_act_pool_fwd_t(const void* y, int io16, const float* stats, const float* gamma, const float* beta, void* out,
                          long long out_fs, int out_ps, int remap_t, int remap_b, int n, int h, int w, int c, int act, int pool,
                          void* stream);
int vad_bn_act_pool_bwd_t(const void* y, int io16, const float* stats, const float* gamma, const float* beta, const void* dout,
                          long long dout_fs, int dout_ps, int remap_t, int remap_b, void* dy, int s2d, float* dgamma, float* dbeta,
                          float* ksums, float* ws, int n, int h, int w, int c, int act, int pool, void* stream);
int vad_lstm_gates_fwd_t(void* z, int io16, const float* c_prev, float* c_out, void* h1, long long h1_fs, int h1_ps,
                         void* h2, long long h2_fs, int h2_ps, int nb, int hw, int hid, void* stream);
int vad_lstm_gates_bwd_t(const void* gates, int io16, const float* c_prev, const float* c, const void* dh1, long long dh1_fs,
                         int dh1_ps, const void* dh2, long long dh2_fs, int dh2_ps, const float* dc_next, void* dz,
                         float* dc_prev, int nb, int hw, int hid, void* stream);
/* Routed first-layer weight gradient of the bf16-tensor step (round 4; csrc/train_ops.hip conv_c3_wgrad_routed_kernel): BatchNorm's
 * backward pass A with `codes` != NULL writes one routing byte per pooled element (argmax position | sign << 2) instead of running
 * pass B; vad_conv_c3_wgrad_routed then forms dW of Conv2d(3 -> 32) + BatchNorm + LeakyReLU + MaxPool2 from the POOLED gradient
 * d(out) ([n, h/2, w/2, 32]: bf16 with io16, bf16 MFMAs throughout; else fp32, T1 on the exact-fp32 MFMA and S in split-fp16), the codes, the input planes and the layer's own weights / statistics (dW = sc (T1 - k1 SX - k2
 * invstd ((W S) + (b - mean) SX)), S the Gram matrix of the input patches) - the dense conv-output gradient is never formed. */
int vad_bn_act_pool_bwd_codes_t(const void* y, int io16, const float* stats, const float* gamma, const float* beta, const void* dout,
                                long long dout_fs, int dout_ps, int remap_t, int remap_b, void* dy, int s2d, float* dgamma,
                                float* dbeta, float* ksums, float* ws, int n, int h, int w, int c, int act, int pool,
                                unsigned char* codes, void* stream);
size_t vad_conv_c3_wgrad_routed_ws_floats(int n, int h);
int vad_conv_c3_wgrad_routed_ok(int h, int w, int cout);
int vad_conv_c3_wgrad_routed(const float* x_nchw, const void* dout, int io16, const unsigned char* codes, const float* w0, const float* b0,
                             const float* stats, const float* gamma, const float* ksums, float* dw, float* ws, int n, int h, int w,
                             int cout, void* stream);
int vad_debug_set_c3_routed(int on);     /* A/B: 0 = pass B + vad_conv_c3_wgrad_t (rounds 1-3); default 1 */
int vad_c3_routed_enabled(void);
int vad_conv_c3_wgrad_t(const float* x_nchw, const void* g, int io16, float* dw, float* ws, int n, int h, int w, int cout,
                        void* stream);
/* grad_mul (a power of two, 1 = none) multiplies every gradient this call emits (din, dpre32, dbias3), not the loss: the
 * split-fp16 training step runs its whole backward on gradients scaled into the fp16 range and unscales the parameter
 * gradients at the end (vad_scale_floats) - exact in fp32 either way. */
int vad_convt_to3_mse_t(const void* in_nhwc, int io16, const float* w_iohw, const float* bias3, const float* x_nchw, float* recon,
                        void* din, void* dpre32, float* loss, float* dbias3, float* ws, int n, int h, int w, float grad_mul,
                        void* stream);
/* p[i] *= mul, i < n. */
int vad_scale_floats(float* p, long long n, float mul, void* stream);
/* Conv2d k1 with `precision`: VAD_PREC_BF16S = bf16 tensors and bf16 operands (weights from vad_train_pack_conv1x1_p with the
 * same precision), anything else = vad_conv1x1 / vad_train_pack_conv1x1. */
int vad_train_pack_conv1x1_p(const float* w_oihw, int cout, int cin, float* fwd, float* dgrad, int precision, void* stream);
int vad_conv1x1_p(const void* in, const float* w_packed, const float* bias, void* out, long long npix, int cin, int cout,
                  int precision, void* stream);
/* First layer (x NCHW fp32 [N,3,H,W]) -> conv3x3(3->Cout) + bias, un-activated, into a bf16 NHWC tensor. */
int vad_conv3x3_c3_bf16(const float* x_nchw, const float* w_packed, const float* bias, void* out_bf16, int n, int h, int w,
                        int cout, void* stream);
/* ... with bf16 MFMA OPERANDS as well (input taps and weights rounded to bf16, nearest even; K = 27 -> 32 = two
 * v_mfma_f32_32x32x16_bf16 per 32 pixels; fp32 accumulation from the bias): what the VAD_PREC_BF16S training step runs. */
int vad_conv3x3_c3_bf16op(const float* x_nchw, const float* w_packed, const float* bias, void* out_bf16, int n, int h, int w,
                          int cout, void* stream);

/* ------------------------------------------------------------------ whole training step (row f-1)
 * Replaces the loop body of train_video.py:50-60 for VideoAutoencoder(in_channels=3, latent_dim, lstm_hidden_dim,
 * lstm_num_layers) (both dims multiples of 32, hidden <= 256; `proj` is the 1x1 conv when they differ): train-mode forward (batch-statistics BatchNorm, running stats updated when `running`
 * is given), nn.MSELoss, and the full backward.  precision VAD_PREC_FP32: exact fp32; VAD_PREC_SPLIT / VAD_PREC_BF16: the 3x3
 * and transposed convolutions (forward + data gradients) use split-fp16 / bf16 operands with fp32 accumulation (bf16: the
 * weight gradients too), the rest (first and last layer, 1x1 data gradients, BatchNorm, gates, loss, Adam, master weights;
 * in split mode also the weight gradients) stays fp32.  VAD_PREC_BF16S: VAD_PREC_BF16 with every activation / gradient tensor
 * of the workspace stored as bf16.  VAD_PREC_WINO: fp32 everywhere, the 3x3 convolutions behind the first layer (forward + data
 * gradients, ConvLSTM gate convolutions included) as Winograd F(2x2,3x3) (csrc/conv_wino.hip).  With more than one ConvLSTM layer
 * and a small batch the layers run as a wavefront on library-owned helper streams, and the weight-gradient GEMMs run on a helper
 * stream beside the BatchNorm / data-gradient chain; both fork from and join `stream` (vad_debug_set_lstm_wavefront(0): serial
 * order, bit-identical).  params / grads: flat fp32 device buffers of vad_vid_train_nparams
 * floats, torch layouts in named_parameters() order (see csrc/train_step.hip); running: vad_vid_train_nstats floats,
 * {running_mean, running_var} per BatchNorm in module order.  x [B,T,3,H,W]; loss: device float[1];
 * recon (nullable) [B,T,3,H,W].  Every gradient is overwritten (no accumulation), so there is no zero_grad.
 * Follow with vad_adam_step on the same flat buffers (after the gradient all-reduce when data-parallel). */
size_t vad_vid_train_nparams(int latent, int hid, int layers);
size_t vad_vid_train_nstats(int latent, int hid, int layers);
size_t vad_vid_train_workspace_bytes(int b, int t, int h, int w, int latent, int hid, int layers);
/* Debug: float offsets of the saved forward buffers inside the training workspace (order documented at the definition in
 * csrc/train_step.hip); returns the number of entries written to out[cap] or a negative VAD_ERR_*. */
/* debug: record the branch decisions (pooling argmax, activation sign) of the following vad_bn_act_pool_bwd calls, one byte
 * per output pixel and channel, consecutively into buf; (NULL, 0) stops.  See csrc/train_ops.hip. */
int vad_debug_set_train_decisions(void* buf, size_t bytes);
size_t vad_debug_train_decisions_used(void);
/* A/B: 0 = the weight gradients of the bf16-tensor mode use the one-channel-per-lane kernel everywhere; 1 = the paired-channel
 * kernel (dword loads, 64 x 64 wave tiles) where cin and ncols are multiples of 64; 2 = its LDS-staged work-group
 * form where ncols is a multiple of 128; 3 (default) = the row-ring kernel for the 3x3 layers with ncols % 64 == 0 and cin % 64 == 0
 * or cin == 32 (every operand row staged once per work-group, csrc/train_ops.hip), 2 elsewhere. */
int vad_debug_set_wgrad_pairs(int on);
/* A/B: 0 = VAD_PREC_SPLIT weight gradients on the exact-fp32 kernel (rounds 2-3); 1 = the per-lane split-fp16 kernel; 2 = its
 * LDS-staged form for the 3x3 layers it takes; 3 (default) = the row-ring kernel for those layers. */
int vad_debug_set_wgrad_split(int on);
/* A/B: 0 = exact-fp32 3x3 weight gradients on the per-wave kernel (rounds 1-3); 1 (default) = the row-ring kernel where it applies. */
int vad_debug_set_wgrad_ring_f32(int on);
/* 1 (default): the BatchNorm forward / backward-apply passes on bf16 tensors take eight channels per thread (16-byte accesses);
 * 0: four, like the fp32 form.  Identical results (every element goes through the same expressions). */
int vad_debug_set_bn_wide(int on);
/* debug / A-B: 0 = the split-fp16 training steps run their backward on unscaled gradients (round 3's form: operands below the
 * fp16 range at large batches); default 1 = scaled by a power of two and unscaled at the end (csrc/train_step.hip). */
int vad_debug_set_split_grad_scale(int on);
int vad_split_grad_scale_enabled(void);
int vad_debug_set_train_stop(int stage);   /* debug: stop vad_vid_train_fwd_bwd after a backward stage (see csrc/train_step.hip) */
int vad_vid_train_debug_layout(int b, int t, int h, int w, int latent, int hid, int layers, long long* out, int cap);
int vad_vid_train_fwd_bwd(const float* x, int b, int t, int h, int w, int latent, int hid, int layers,
                          const float* params, float* grads, float* running, void* workspace, size_t workspace_bytes,
                          int precision, float* loss, float* recon, void* stream);

/* Image autoencoder counterpart (train.py:28-52; not a SURVEY section 8 row): ConvAutoencoder(in_channels=3, latent_dim), x
 * [N,3,H,W]; loss_kind 0 = nn.MSELoss (train.py default), 1 = SSIMLoss(window_size), 2 = CombinedLoss(alpha, window_size)
 * (train.py:149-158).  Same flat-buffer conventions and arithmetic modes as the video step. */
size_t vad_img_train_nparams(int latent);
size_t vad_img_train_nstats(int latent);
size_t vad_img_train_workspace_bytes(int n, int h, int w, int latent);
int vad_img_train_fwd_bwd(const float* x, int n, int h, int w, int latent, const float* params, float* grads,
                          float* running, void* workspace, size_t workspace_bytes, int loss_kind, float alpha,
                          int window_size, int precision, float* loss, float* recon, void* stream);

/* Synthetic frames on device, bit-identical to synth.frames() (numpy): NCHW fp32 in [-1,1].  anomalies: 0 = none,
 * 1 = labelled frames carry a saturated 32 x 32 patch, k >= 2 = a k x k patch. */
int vad_synth_frames(float* out_nchw, unsigned long long seed, long long first_frame, long long n,
                     int c, int h, int w, int anomalies, void* stream);

/* ------------------------------------------------------------------ whole-model scoring
 * Image autoencoder.  Replaces ConvAutoencoder.forward / get_latent / get_reconstruction_error
 * (models/autoencoder.py:181-221) as driven by evaluate.compute_auroc (evaluate.py:56-64).
 * params: VAD_IMG_NPARAMS host pointers in state_dict order with num_batches_tracked removed. */
#define VAD_IMG_NPARAMS 92
/* latent_dim / lstm_hidden_dim: ANY value in [1, VAD_MAX_WIDTH], as the reference's constructors take
 * (models/autoencoder.py:161, models/video_autoencoder.py:290-296; train_video.py:304-326 exposes them as flags).  The
 * kernels tile channels by 32 (64 hidden channels per ConvLSTM block); the packers zero-pad other widths (zero weights and
 * bias in, zero weights out: a padded channel stays exactly 0 through every layer, results are those of the unpadded
 * network).  Every entry point takes the REAL dimensions. */
#define VAD_MAX_WIDTH 4096
/* The packed blob starts with a 16-byte header {magic "VADB", tag = abi<<16 | precision<<8 | model kind, dims}; the
 * layers follow.  Pack with the precision the blob will be launched with: the kernel that finalises the scores compares
 * the tag with the launch's `precision` on the device and returns NaN scores on a mismatch (never a silent wrong number).
 * vad_blob_precision reads the mode back from a HOST copy of a blob (negative VAD_ERR_ARG if it is not one). */
size_t vad_img_packed_floats(int in_ch, int latent);
int vad_img_pack(const float* const* params, int nparams, int in_ch, int latent, int precision, float* packed_host);
int vad_blob_precision(const float* packed_host);
size_t vad_img_workspace_bytes(int chunk, int h, int w, int latent);
/* x NCHW [B,3,H,W] on device.  Frames are processed in chunks of `chunk` through `workspace`.
 * Outputs (device, any may be NULL): scores [B]; errmap [B,H,W]; recon NCHW [B,3,H,W];
 * latent NCHW [B,latent,H/16,W/16].  This short form is float input + VAD_PREC_FP32 (blob packed likewise);
 * vad_img_score_x takes the input format and the arithmetic mode. */
int vad_img_score(const float* x_nchw, long long b, int h, int w, int latent,
                  const float* packed_dev, void* workspace, size_t workspace_bytes, int chunk,
                  float* scores, float* errmap, float* recon_nchw, float* latent_nchw, void* stream);

/* Video ConvLSTM autoencoder.  Replaces VideoAutoencoder.forward / get_reconstruction_error
 * (models/video_autoencoder.py:329-384) as driven by evaluate_video.py:138-154,346-352.
 * params: host pointers in state_dict order with num_batches_tracked removed:
 *   4 x (conv w,b, bn g,b,m,v) ; layers x (cell w,b) ; [proj w,b] ; 3 x (convT w,b, bn g,b,m,v) ; convT w,b */
int vad_vid_nparams(int layers, int has_proj);
size_t vad_vid_packed_floats(int latent, int hid, int layers);
int vad_vid_pack(const float* const* params, int nparams, int latent, int hid, int layers, int precision, float* packed_host);
size_t vad_vid_workspace_bytes(int chunk_clips, int t, int h, int w, int latent, int hid, int layers);
/* x [B,T,3,H,W].  Outputs (any may be NULL): seq_scores [B]; frame_scores [B,T];
 * errmap [B,T,H,W]; recon [B,T,3,H,W].  Asynchronous on `stream`; launch groups smaller than one work-group per CU run the
 * ConvLSTM layers as a wavefront on library-owned helper streams that fork from and join `stream` (vad_debug_set_lstm_wavefront). */
int vad_vid_score(const float* x, long long b, int t, int h, int w, int latent, int hid, int layers,
                  const float* packed_dev, void* workspace, size_t workspace_bytes, int chunk_clips,
                  float* seq_scores, float* frame_scores, float* errmap, float* recon, void* stream);

/* Developer switches for A/B timing in one process (process-wide, debug only; the library never writes them itself).
 * Bit 0: 1 = persistent work-groups with register prefetch of the next stage (default), 0 = one tile per work-group
 * (bit-identical results).  Bit 1: pricing runs whose results are NOT valid - exact kernels drop their epilogue stores,
 * split kernels read no weights.  Bit 2: alternative cout-64 tiling.  Bit 3: never use the small-grid (16x16x4) ConvLSTM
 * kernel, bit 4: always use it, bit 5: never compute the ConvLSTM x halves ahead of the recurrence, bit 6: never use the
 * gate-split form of the small-grid kernel (one gate per wave, for the smallest grids), bit 7: use its 8-wave form wherever the small-grid
 * kernel would run (bit-identical results either way). */
int vad_debug_set_conv_variant(int variant);
/* 0 = run the ConvLSTM layers strictly one after the other on the caller's stream; 1 (default) = small launch groups run
 * them as a wavefront: layer l step t on a library-owned helper stream as soon as layer l-1 step t is done (fork / join by
 * events on `stream`; the helper streams are created once per thread and device on the first such call).  Same results. */
int vad_debug_set_lstm_wavefront(int on);
/* Frames per dec4.0 -> scoring-tail sub-group inside vad_img_score (0 = whole launch group). */
int vad_debug_set_tail_group(int frames);
/* A/B: 0 = dec4.0 and the scoring tail of the image model as two launches (round-2 path), 1 = the fused kernel (default) */
int vad_debug_set_dec4_fused(int on);
/* rows of the dec3.3 map per work-group band of the fused kernel (0 = chosen from the batch size); results do not depend on it */
int vad_debug_set_dec4_band(int rows);

/* Row f-3 — raw-frame ingest.  The *_x forms take the original frames in either format:
 *   VAD_X_F32_NCHW (0): float32 [N,3,H,W] already normalised to [-1,1] (same as the plain entry points);
 *   VAD_X_U8_NHWC  (1): uint8 [N,H,W,3] as decoded from an image / video file.  ToTensor + Normalize(0.5, 0.5)
 *                       (reference utils/dataset.py:65-70, utils/video_dataset.py:62-66) is applied inside the first
 *                       convolution's staging load and inside the scoring tail, bit-identically to the fp32 path, so
 *                       the normalised fp32 frames (4x the bytes) never exist in memory.
 * `precision` is the arithmetic mode the blob was packed for (VAD_PREC_*).  All other arguments and outputs are those of
 * vad_img_score / vad_vid_score / vad_vid_score_windows (which are the float-input, VAD_PREC_FP32 short forms). */
#define VAD_X_F32_NCHW 0
#define VAD_X_U8_NHWC 1
int vad_img_score_x(const void* x, int x_format, int precision, long long b, int h, int w, int latent, const float* packed_dev,
                    void* workspace, size_t workspace_bytes, int chunk, float* scores, float* errmap, float* recon_nchw,
                    float* latent_nchw, void* stream);
int vad_vid_score_x(const void* x, int x_format, int precision, long long b, int t, int h, int w, int latent, int hid, int layers,
                    const float* packed_dev, void* workspace, size_t workspace_bytes, int chunk_clips, float* seq_scores,
                    float* frame_scores, float* errmap, float* recon, void* stream);
int vad_vid_score_windows_x(const void* frames, int x_format, int precision, long long nframes, int t, int stride, int h, int w,
                            int latent, int hid, int layers, const float* packed_dev, void* workspace,
                            size_t workspace_bytes, int chunk_windows, float* seq_scores, float* frame_scores,
                            float* errmap, float* recon, void* stream);

/* Dense sliding-window scoring of ONE video: window k = frames [k*stride, k*stride + T), 0 < stride <= T,
 * vad_vid_num_windows(F, T, stride) = (F - T) / stride + 1 windows.  Same results as scoring every window as its own
 * clip with vad_vid_score (what the reference does: evaluate_video.py:322-352 builds VideoFileDataset(sequence_length,
 * stride=1) and runs 3 forwards per window), but every frame is encoded once instead of once per window containing it.
 * frames [F,3,H,W].  Outputs (any may be NULL): seq [NW]; frame [NW,T]; errmap [NW,T,H,W]; recon [NW,T,3,H,W]. */
long long vad_vid_num_windows(long long frames, int t, int stride);
size_t vad_vid_windows_workspace_bytes(int chunk_windows, int t, int stride, int h, int w, int latent, int hid, int layers);
int vad_vid_score_windows(const float* frames, long long nframes, int t, int stride, int h, int w, int latent, int hid,
                          int layers, const float* packed_dev, void* workspace, size_t workspace_bytes, int chunk_windows,
                          float* seq_scores, float* frame_scores, float* errmap, float* recon, void* stream);

/* Models with in_channels > 3 (round 4).  ConvAutoencoder(in_channels=...) / VideoAutoencoder(in_channels=...) take any width
 * (models/autoencoder.py:161, models/video_autoencoder.py:290-296; every reference call site passes 3: evaluate.py:35,
 * evaluate_video.py:99-104).  The `_c` forms below take in_ch in [3, VAD_MAX_IN_CHANNELS]; with in_ch == 3 they ARE the forms
 * above.  A wider model runs its first and last layer on the generic kernels over planes zero-padded to 32 channels
 * (csrc/wide_io.hip: one NCHW -> padded-NHWC copy on the way in, Tanh + squared error on the way out); blobs come from
 * vad_img_pack(..., in_ch, ...) / vad_vid_pack_c, inputs are float NCHW [.., in_ch, H, W] (uint8 frames are 3-channel
 * images), recon comes back with in_ch planes, scores and error maps average over in_ch planes.  1- and 2-channel models: pack
 * and score them as 3-channel models with zero weights on the extra planes (what the Python layer does). */
#define VAD_MAX_IN_CHANNELS 32
size_t vad_img_workspace_bytes_c(int chunk, int h, int w, int latent, int in_ch);
int vad_img_score_c(const void* x, int x_format, int precision, int in_ch, long long b, int h, int w, int latent,
                    const float* packed_dev, void* workspace, size_t workspace_bytes, int chunk, float* scores, float* errmap,
                    float* recon_nchw, float* latent_nchw, void* stream);
size_t vad_vid_packed_floats_c(int in_ch, int latent, int hid, int layers);
int vad_vid_pack_c(const float* const* params, int nparams, int in_ch, int latent, int hid, int layers, int precision,
                   float* packed_host);
size_t vad_vid_workspace_bytes_c(int chunk_clips, int t, int h, int w, int latent, int hid, int layers, int in_ch);
int vad_vid_score_c(const void* x, int x_format, int precision, int in_ch, long long b, int t, int h, int w, int latent, int hid,
                    int layers, const float* packed_dev, void* workspace, size_t workspace_bytes, int chunk_clips,
                    float* seq_scores, float* frame_scores, float* errmap, float* recon, void* stream);
size_t vad_vid_windows_workspace_bytes_c(int chunk_windows, int t, int stride, int h, int w, int latent, int hid, int layers,
                                         int in_ch);
int vad_vid_score_windows_c(const void* frames, int x_format, int precision, int in_ch, long long nframes, int t, int stride,
                            int h, int w, int latent, int hid, int layers, const float* packed_dev, void* workspace,
                            size_t workspace_bytes, int chunk_windows, float* seq_scores, float* frame_scores, float* errmap,
                            float* recon, void* stream);

/* ------------------------------------------------------------------ hipGraph capture / replay of a scoring call
 * vad_graph_begin(stream); <one vad_img_score* / vad_vid_score* call on `stream`>; vad_graph_end(stream, &exec) captures
 * the call's launch sequence (kernels, and the fork / join with the library's helper streams) into an instantiated
 * hipGraph; vad_graph_launch(exec, stream) replays it asynchronously with the SAME device pointers (keep input, output and
 * workspace buffers alive and write new frames into the same input buffer).  Run the call once eagerly first: the first
 * call of a thread creates helper streams and queries occupancy, which must not happen inside a capture.  Per-layer timing
 * is skipped while capturing.  Results are bit-identical to the eager call. */
int vad_graph_begin(void* stream);
int vad_graph_end(void* stream, void** exec_out);
int vad_graph_launch(void* exec, void* stream);
int vad_graph_destroy(void* exec);

/* ------------------------------------------------------------------ per-layer timing
 * When enabled, the model-level calls bracket every layer launch with hipEvents on `stream`.
 * vad_prof_read synchronises the events and returns the accumulated ms and launch count per
 * layer slot since the last vad_prof_reset.  The record list is mutex-guarded: threads may score concurrently while
 * profiling is on (their records are pooled). */
#define VAD_PROF_SLOTS 32
int vad_prof_enable(int on);
int vad_prof_reset(void);
int vad_prof_read(float* ms /*[VAD_PROF_SLOTS]*/, int* launches /*[VAD_PROF_SLOTS]*/);
const char* vad_prof_slot_name(int model /*0 img, 1 vid, 2 video training step (kernel groups)*/, int slot);

#ifdef __cplusplus
}
#endif
#endif
