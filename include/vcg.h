/*
 * vcg.h -- C ABI of the MI355X (gfx950) hot-path library  libvcg_hip.so
 *
 * Scope: the generator / discriminator forward+backward train step of
 * kjedrzejewski/video-cycle_gan-upscaling.  The reference has NO FFI / plugin
 * boundary of its own (it is pure Keras); every entry point below replaces one third-party Keras/TF
 * primitive at the call site quoted next to it (paths relative to the reference root).
 *
 * Conventions
 *   - all tensors are fp32, contiguous, NCHW on the device; kernels keep Keras' layouts:
 *       Conv2D          kernel (kh,kw,in,out)  "HWIO"   + its per-tap transpose (kh,kw,out,in) "HWOI"
 *       Conv2DTranspose kernel (kh,kw,out,in)  "HWOI"   + its per-tap transpose "HWIO"
 *       Dense           kernel (in,out)
 *   - the library never allocates, frees or synchronises: the caller passes every buffer and a
 *     workspace (size from the *_workspace_bytes query); every call is asynchronous on `stream`
 *     and hipGraph-capturable;
 *   - return value: 0 ok; <0 invalid argument (VCG_E_*); >0 a hipError_t;
 *   - entry points are re-entrant and stateless.
 */
#ifndef VCG_H
#define VCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vcg_stream_t; /* a hipStream_t */

enum {
    VCG_OK = 0,
    VCG_E_NULL = -1,        /* required pointer is NULL              */
    VCG_E_SHAPE = -2,       /* inconsistent / unsupported dimensions */
    VCG_E_UNSUPPORTED = -3, /* kernel size / stride not instantiated */
    VCG_E_WORKSPACE = -4    /* workspace too small                   */
};

enum { VCG_ACT_NONE = 0, VCG_ACT_LRELU = 1, VCG_ACT_PRELU = 2, VCG_ACT_TANH = 3 };
enum { VCG_NORM_BATCH = 0, VCG_NORM_INSTANCE = 1 };
enum { VCG_LOSS_MSE = 0, VCG_LOSS_MAE = 1 };
/* discriminator output activation (upscaling/upscaler/model.py:885-892) == GanLosses.loss_activation (:172-181) */
enum { VCG_HEAD_NONE = 0, VCG_HEAD_SIGMOID = 1, VCG_HEAD_LOGSIGM = 2, VCG_HEAD_TANH = 3, VCG_HEAD_BILOG = 4 };

/* A Keras Conv2D / Conv2DTranspose call site.  For Conv2D  (n,cin,h,w) -> (n,cout,oh,ow);
 * pad_top/pad_left are the TF-SAME "before" pads (bottom/right follow from oh/ow).
 * For Conv2DTranspose (n,cin,h,w) -> (n,cout,oh,ow) with oh<=h*stride; pad_top/pad_left are the
 * crop-before amounts floor((k-s)/2) of the full transposed convolution. */
typedef struct vcg_conv_desc {
    int32_t n, cin, h, w;
    int32_t cout, oh, ow;
    int32_t kh, kw, stride;
    int32_t pad_top, pad_left;
} vcg_conv_desc;

/* Fused output stage: y = act(acc + bias) + residual */
typedef struct vcg_epilogue {
    const float* bias;        /* [cout] or NULL                        */
    int32_t act;              /* VCG_ACT_*                             */
    float act_alpha;          /* LeakyReLU slope                       */
    const float* prelu_alpha; /* [cout] per-channel slope (ACT_PRELU)  */
    const float* residual;    /* same shape as the output, or NULL     */
} vcg_epilogue;

const char* vcg_version(void);
const char* vcg_error_string(int code);

/* ---- Conv2D: keras.layers.Conv2D at upscaling/upscaler/model.py:19,22,275,283,290 (generator),
 *      :839-871 / :904-936 (discriminators), :65 (downsampling_block) -------------------------- */
int vcg_conv2d_fwd(const vcg_conv_desc* d, const float* x, const float* w_hwio, float* y,
                   const vcg_epilogue* ep, vcg_stream_t stream);
/* vcg_conv2d_fwd (+ bias, no activation) whose epilogue also leaves the statistics of the BatchNormalization / instance norm that follows
 * (upscaling/upscaler/model.py:19-25, 284, 840): per output tile and channel the sum of (y - bias) and of its square, stats fp32
 * [n][records per image][2][cout] with records = vcg_conv2d_stats_records(d, stats_mode) (stats_mode: 1 = VCG_STATS_BATCH -> n * tiles,
 * 2 = VCG_STATS_INSTANCE -> tiles per image); feed them to vcg_norm_finalize_partials_shifted(kshift = bias).  3x3 / 4x4, stride 1-2, more than
 * 3 input channels; a negative record count / VCG_E_UNSUPPORTED otherwise (run vcg_norm_stats on the output). */
int vcg_conv2d_stats_records(const vcg_conv_desc* d, int stats_mode);
int vcg_conv2d_fwd_stats(const vcg_conv_desc* d, const float* x, const float* w_hwio, const float* bias, float* y, float* stats,
                         vcg_stream_t stream);
/* dx = dL/dx given dy; `residual` (optional, shape of dx) is added: dx = dgrad + residual */
int vcg_conv2d_dgrad(const vcg_conv_desc* d, const float* dy, const float* w_hwio, const float* w_hwoi,
                     float* dx, const float* residual, vcg_stream_t stream);
size_t vcg_conv2d_wgrad_workspace_bytes(const vcg_conv_desc* d);
/* dw_hwio (overwritten), dbias (overwritten, may be NULL) */
int vcg_conv2d_wgrad(const vcg_conv_desc* d, const float* x, const float* dy, float* dw_hwio, float* dbias,
                     void* ws, size_t ws_bytes, vcg_stream_t stream);

/* ---- Conv2DTranspose(strides=2, padding='same'): upscaling/upscaler/model.py:72 ---------------- */
int vcg_conv_transpose2d_fwd(const vcg_conv_desc* d, const float* x, const float* w_hwio_t, float* y,
                             const vcg_epilogue* ep, vcg_stream_t stream);
int vcg_conv_transpose2d_dgrad(const vcg_conv_desc* d, const float* dy, const float* w_hwoi, float* dx,
                               const float* residual, vcg_stream_t stream);
size_t vcg_conv_transpose2d_wgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv_transpose2d_wgrad(const vcg_conv_desc* d, const float* x, const float* dy, float* dw_hwoi,
                               float* dbias, void* ws, size_t ws_bytes, vcg_stream_t stream);

/* per-tap transpose of a kernel: (taps, a, b) -> (taps, b, a) */
int vcg_kernel_transpose(const float* src, float* dst, int taps, int a, int b, vcg_stream_t stream);

/* ---- BatchNormalization (+PReLU/LeakyReLU, +Add): model.py:20-25,284-285,840-841;
 *      instance norm is the north_star extension ------------------ */
/* per-channel (batch) or per-(n,c) (instance) mean and biased variance of x[n,c,hw].
 * mean/var: [c] (batch) or [n*c] (instance).  ws: vcg_norm_stats_workspace_bytes. */
size_t vcg_norm_stats_workspace_bytes(int n, int c, int hw, int mode);
int vcg_norm_stats(const float* x, int n, int c, int hw, int mode, float* mean, float* var,
                   void* ws, size_t ws_bytes, vcg_stream_t stream);
/* scale = gamma*rsqrt(var+eps), shift = beta - mean*scale ; gamma/beta may be NULL (1/0);
 * moving_* (optional): moving = moving*momentum + batch*(1-momentum), var corrected by
 * count/(count-1) when unbiased_count>1.  count entries = c or n*c. */
int vcg_norm_finalize(const float* mean, const float* var, const float* gamma, const float* beta, int c,
                      int rows, float eps, float* scale, float* shift, float* invstd,
                      float* moving_mean, float* moving_var, float momentum, int unbiased_count,
                      vcg_stream_t stream);
/* out[ch] = scale * sum over rec of part[rec][ch] (part fp32 [nrec][c]), in double and in a fixed order: the tail of every kernel that
 * leaves per-workgroup records instead of using atomics */
int vcg_sum_records(const float* part, int nrec, int c, float scale, float* out, vcg_stream_t stream);
/* inference-mode BatchNormalization folded into the epilogue of the convolution in front of it (model.py:19-20 with learning phase 0):
 * scale = gamma * rsqrt(moving_var + eps), shift = (bias - moving_mean) * scale + beta; bias / gamma / beta may be NULL (0 / 1 / 0) */
int vcg_bn_fold(const float* bias, const float* moving_mean, const float* moving_var, const float* gamma, const float* beta, int c, float eps,
                float* scale, float* shift, vcg_stream_t stream);
/* the same for `count` (<= 48) pairs in one launch: HOST arrays of device pointers (bias / gamma / beta entries may be NULL); scale, shift [count][c] */
int vcg_bn_fold_batch(const float* const* bias, const float* const* moving_mean, const float* const* moving_var, const float* const* gamma,
                      const float* const* beta, int count, int c, float eps, float* scale, float* shift, vcg_stream_t stream);
/* the same from partial records written by a convolution's epilogue (vcg_epilogue_bf16.stats, vcg_conv2d_nhwc_bf16_fwd_stats):
 * part fp32 [groups][nrec][2][c] = per-record sums and sums of squares over `count` values per group and channel in all; groups = 1
 * (batch statistics; only then are the moving averages updated) or n (instance norm).  Writes mean, scale, shift, invstd [groups*c].
 * One launch in place of vcg_norm_stats* + vcg_norm_finalize; the records are added in a fixed order (deterministic). */
int vcg_norm_finalize_partials(const float* part, int nrec, int groups, int c, double count, const float* gamma, const float* beta,
                               float eps, float* mean, float* scale, float* shift, float* invstd, float* moving_mean,
                               float* moving_var, float momentum, int unbiased_count, vcg_stream_t stream);
/* the same for records that hold sums of (x - kshift[ch]) and of its square (vcg_conv2d_fwd_stats: kshift = the convolution's bias): the
 * variance is that of the shifted values, the mean gets the shift back */
int vcg_norm_finalize_partials_shifted(const float* part, int nrec, int groups, int c, double count, const float* kshift, const float* gamma,
                                       const float* beta, float eps, float* mean, float* scale, float* shift, float* invstd, float* moving_mean,
                                       float* moving_var, float momentum, int unbiased_count, vcg_stream_t stream);
/* y = act(x*scale + shift) + residual ; scale/shift indexed [c] (rows==1) or [n*c] */
int vcg_norm_act_fwd(const float* x, int n, int c, int hw, const float* scale, const float* shift,
                     int per_sample, int act, float act_alpha, const float* prelu_alpha,
                     const float* residual, float* y, vcg_stream_t stream);
/* backward of y = act(z), z = gamma*xhat+beta, xhat=(x-mean)*invstd:
 *   pass 1 (reduce):  sum_dz, sum_dz_xhat per channel (or per (n,c)), dalpha per channel
 *   pass 2 (apply):   dx = scale*(dz - sum_dz/M - xhat*sum_dz_xhat/M)      (training statistics)
 *                or   dx = scale*dz                                          (frozen statistics)
 * dgamma[c] = sum_dz_xhat, dbeta[c] = sum_dz (summed over n for instance mode). */
size_t vcg_norm_act_bwd_workspace_bytes(int n, int c, int hw, int mode);
int vcg_norm_act_bwd(const float* x, const float* dy, int n, int c, int hw, int mode,
                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                     int act, float act_alpha, const float* prelu_alpha, int use_batch_stats,
                     float* dx, float* dgamma, float* dbeta, float* dprelu_alpha,
                     void* ws, size_t ws_bytes, vcg_stream_t stream);

/* ---- plain activations: PReLU after 'initial/conv' (model.py:276), LeakyReLU (:73), tanh (:291) */
/* dx = dy * act'(.) where the derivative is evaluated from the saved OUTPUT y (tanh, lrelu) or the
 * saved INPUT x (prelu).  dprelu_alpha[c] (optional) = sum dy*min(x,0); dsum[c] (optional) = sum dx over
 * (n,hw) -- the bias gradient of the convolution in front, produced in the same pass. */
size_t vcg_act_bwd_workspace_bytes(int n, int c, int hw);
int vcg_act_bwd(const float* saved, const float* dy, int n, int c, int hw, int act, float act_alpha,
                const float* prelu_alpha, float* dx, float* dprelu_alpha, float* dsum, void* ws, size_t ws_bytes,
                vcg_stream_t stream);
/* out[c] = sum over n,hw of x */
size_t vcg_channel_sum_workspace_bytes(int n, int c, int hw);
int vcg_channel_sum(const float* x, int n, int c, int hw, float* out, void* ws, size_t ws_bytes,
                    vcg_stream_t stream);

/* ---- Dense: model.py:876,880,884 -------------------------------------------------------------- */
int vcg_dense_fwd(const float* x, const float* w_io, const float* bias, float* y, int batch, int in,
                  int out, vcg_stream_t stream);
int vcg_dense_dgrad(const float* dy, const float* w_io, float* dx, int batch, int in, int out,
                    vcg_stream_t stream);
int vcg_dense_wgrad(const float* x, const float* dy, float* dw_io, float* dbias, int batch, int in,
                    int out, vcg_stream_t stream);

/* ---- losses: model.py:159-160, 215-261; pixel term of model.py:137,157 ------------------------ */
/* out[0] = mean(x) over count elements (deterministic two-stage) */
size_t vcg_mean_reduce_workspace_bytes(size_t count);
int vcg_mean_reduce(const float* x, size_t count, float* out, void* ws, size_t ws_bytes, vcg_stream_t stream);
/* content loss value (out[0]) and gradient d/dpred scaled by grad_scale/count */
int vcg_pixel_loss(const float* pred, const float* target, size_t count, int kind, float grad_scale,
                   float* out, float* dpred, void* ws, size_t ws_bytes, vcg_stream_t stream);
/* Discriminator output activation: Activation('sigmoid') / Lambda(log(sigmoid)) / Activation('tanh') /
 * Lambda(x/(1+|x|)*log(|x|+2)) after Dense_3 (upscaling/upscaler/model.py:885-892, 950-957, 1001-1008).
 * fwd: y = act(z); bwd: dz = dy * act'(z) from the saved PRE-activation z.  kind: VCG_HEAD_*. */
int vcg_head_act_fwd(const float* z, float* y, size_t count, int kind, vcg_stream_t stream);
int vcg_head_act_bwd(const float* z, const float* dy, float* dz, size_t count, int kind, vcg_stream_t stream);
/* GAN losses evaluated on the device from two mean(D(.)) scalars (model.py:220-233 Wasserstein: kind = VCG_HEAD_NONE;
 * model.py:244-259 relativistic: loss_activation(mean_a - mean_b)):
 *   delta = (mean_a[0] - (mean_b ? mean_b[0] : 0)) * mean_scale        (mean_scale = 1/ranks when the scalars hold an
 *                                                                       all-reduce SUM of per-rank means, else 1)
 *   loss_out[0] = act(delta)                 (optional)
 *   da[0..na) = act'(delta) * ga ,  db[0..nb) = act'(delta) * gb       -- the broadcast gradients dL/dD(.) of the two
 *                                                                       batches the means were taken over
 * No host read: the train step stays capturable in a hipGraph (train_gan3.py:353-354 returns the loss to the host
 * after the step). */
int vcg_gan_loss(const float* mean_a, const float* mean_b, float mean_scale, int kind, float* loss_out, float* da, size_t na,
                 float ga, float* db, size_t nb, float gb, vcg_stream_t stream);
/* Attention gates of make_upscaler_attention (upscaling/upscaler/model.py:34-36, 86-89): Activation('sigmoid') + Multiply.
 * fwd: y = sigmoid(a) * m;  bwd: da = dy * m * sigmoid'(a) (gradient wrt the PRE-sigmoid tensor), dm = dy * sigmoid(a) */
int vcg_sigmoid_gate_fwd(const float* a, const float* m, float* y, size_t count, vcg_stream_t stream);
int vcg_sigmoid_gate_bwd(const float* a, const float* m, const float* dy, float* da, float* dm, size_t count, vcg_stream_t stream);
/* Lambda(atanh(0.99999 * x)) applied to the network input (model.py:94): y = atanh(scale * x); no gradient (x is data) */
int vcg_atanh_scale(const float* x, float* y, size_t count, float scale, vcg_stream_t stream);
/* zero insertion dst[plane][y*s][x*s] = src[plane][y][x] (dst: (h-1)*s+1 x (w-1)*s+1, zeros elsewhere).  The data gradient of a
 * stride-3 convolution (make_discriminator_sparse_512, upscaling/upscaler/model.py:971-987) = vcg_conv2d_dgrad with stride 1 over
 * the dilated gradient. */
int vcg_dilate2d(const float* src, float* dst, size_t planes, int h, int w, int stride, vcg_stream_t stream);
/* Lambda(K.resize_images(x, f, f, 'channels_last', 'nearest' | 'bilinear')) of the skip-connection / attention / U-Net-ish
 * generators (upscaling/upscaler/model.py:80-81,352,705,786) with Keras 2.2.x on TF 1.14 semantics: tf.image.resize_nearest_neighbor /
 * resize_bilinear, align_corners=False, no half-pixel centres (source coordinate = destination / f).  dst: planes x (h*f) x (w*f).
 * Forward only: every use in the reference resizes the network INPUT, which is data. */
int vcg_resize2d(const float* src, float* dst, size_t planes, int h, int w, int factor, int bilinear, vcg_stream_t stream);
/* Cropping2D(((top, bottom), (left, right))) (model.py:552,563,626): dst[pl][y][x] = src[pl][y + top][x + left], dst oh x ow */
int vcg_crop2d(const float* src, float* dst, size_t planes, int h, int w, int top, int left, int oh, int ow, vcg_stream_t stream);
/* its gradient: dst (oh x ow) holds src (h x w) at offset (top, left), zeros elsewhere */
int vcg_pad2d(const float* src, float* dst, size_t planes, int h, int w, int top, int left, int oh, int ow, vcg_stream_t stream);
/* Concatenate(axis=3) (model.py:353,406,435,553) and its gradient in NCHW = block copies of channels:
 * dst[n][c_dst_off + c][hw] = src[n][c_src_off + c][hw], c < c_count (src has c_src channels per image, dst c_dst) */
int vcg_copy_channels(const float* src, float* dst, int n, int c_src, int c_src_off, int c_dst, int c_dst_off, int c_count, size_t hw,
                      vcg_stream_t stream);
/* Dropout(rate) in the learning phase (model.py:510,519,528): keep = u >= rate with u a counter-based hash of (seed, *step, element
 * index) -- *step is read on the device so a recorded hipGraph draws a new mask per replay (vcg_counter_inc advances it);
 * y = keep ? x / (1 - rate) : 0 (tf.nn.dropout); the mask (1 byte per element) is written for the backward: dx = dy * mask / (1 - rate) */
int vcg_dropout_fwd(const float* x, float* y, unsigned char* mask, size_t count, float rate, unsigned long long seed,
                    const unsigned long long* step, vcg_stream_t stream);
int vcg_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, size_t count, float rate, vcg_stream_t stream);
int vcg_counter_inc(unsigned long long* counter, vcg_stream_t stream);
/* y = value everywhere (broadcast gradient of a mean) */
int vcg_fill(float* y, size_t count, float value, vcg_stream_t stream);
/* y = a*x + b*y */
int vcg_axpby(const float* x, float* y, size_t count, float a, float b, vcg_stream_t stream);

/* MaxPooling2D((2,2), strides (2,2), 'valid') of the VGG19 feature extractor behind VGG_LOSS / VGG_MSE_LOSS /
 * VGG_MAE_LOSS (upscaling/upscaler/model.py:101-157; keras.applications.VGG19 block{1..4}_pool).  fp32 NCHW;
 * y: [n,c,h/2,w/2] (floor).  bwd: gradient to the first maximal element of each window (row-major), x = forward input. */
int vcg_maxpool2x2_fwd(const float* x, float* y, int n, int c, int h, int w, vcg_stream_t stream);
int vcg_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int n, int c, int h, int w, vcg_stream_t stream);

/* ---- Adam(): keras.optimizers.Adam defaults, model.py:1026,1066,1130 -------------------------- */
/* multi-tensor update over one flat parameter buffer; lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed
 * by the caller; p -= lr_t * m / (sqrt(v) + eps).  The gradient enters as g * grad_scale: 1 for a single process,
 * 1/ranks when g holds the all-reduce SUM of the data-parallel replicas' buckets (no separate averaging pass). */
int vcg_adam_keras_multi(float* p, const float* g, float* m, float* v, size_t count, float lr_t,
                         float beta_1, float beta_2, float eps, float grad_scale, vcg_stream_t stream);

/* same update with the iteration count kept on the device, so a hipGraph that captured the call replays correctly
 * step after step.  t_dev points to TWO 32-bit words: [0] the int32 iteration count (uses t = t_dev[0] + 1 for the
 * bias correction, then increments it), [1] scratch for lr_t (evaluated once per call, in double, by one thread) */
int vcg_adam_keras_multi_dev(float* p, const float* g, float* m, float* v, size_t count, float lr,
                             float beta_1, float beta_2, float eps, float grad_scale, int* t_dev, vcg_stream_t stream);

/* ---- frame edge: upscaling/upscaler/data.py:253-270 ------------------------------------------- */
/* uint8 NHWC -> fp32 NCHW, v/127.5 - 1 */
int vcg_frames_u8_to_nchw(const uint8_t* src, float* dst, int n, int h, int w, int c, vcg_stream_t stream);
/* fp32 NCHW -> uint8 NHWC, uint8(around((a+1)*127.5)) */
int vcg_nchw_to_frames_u8(const float* src, uint8_t* dst, int n, int h, int w, int c, vcg_stream_t stream);
int vcg_nhwc_to_nchw(const float* src, float* dst, int n, int h, int w, int c, vcg_stream_t stream);
int vcg_nchw_to_nhwc(const float* src, float* dst, int n, int h, int w, int c, vcg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * bf16-storage path (BASELINE.json configs C3-C5): activations bf16 NHWC, weights bf16 packed
 * [tap][out-channel][in-channel], fp32 accumulation (v_mfma_f32_32x32x16_bf16) and fp32 epilogue arithmetic.
 * ------------------------------------------------------------------------------------------------------------ */

/* y = act(acc * scale[c] + shift[c]) + residual.  scale/shift: fp32 per out-channel or NULL (1 / 0) -- the
 * inference-mode BatchNormalization folded behind the convolution (Keras: upscaling/upscaler/model.py:20,23,284;
 * shift also carries the Conv2D bias); act: VCG_ACT_NONE / _LRELU (act_alpha) / _PRELU (prelu_alpha[c], model.py:21);
 * residual: bf16 NHWC tensor of the output's shape or NULL (the block's Add, model.py:25,285). */
typedef struct vcg_epilogue_bf16 {
    const void* scale;
    const void* shift;
    int32_t act;
    float act_alpha;
    const void* prelu_alpha;
    const void* residual;
    /* statistics of the stored output for the normalisation BEHIND the convolution (training-mode BatchNormalization, model.py:20,23,284;
     * instance norm): stats_mode VCG_STATS_NONE, or VCG_STATS_BATCH / VCG_STATS_INSTANCE with `stats` = fp32
     * [groups][records][2][cout] (groups = 1 / n; records = vcg_conv2d_bf16_stats_records): per-record sums and sums of squares that
     * vcg_norm_finalize_partials adds up in a fixed order.  Replaces the vcg_norm_stats_bf16 pass over the output. */
    void* stats;
    int32_t stats_mode;
} vcg_epilogue_bf16;
#define VCG_STATS_NONE 0
#define VCG_STATS_BATCH 1
#define VCG_STATS_INSTANCE 2

/* fp32 kernel -> packed bf16.  out[tap'][i][j] (j contiguous) = transpose ? w[tap][j][i] : w[tap][i][j], with
 * tap' = flip ? taps-1-tap : tap.  Conv2D forward from Keras' (kh,kw,in,out): a=out, b=in, transpose=1, flip=0;
 * its data gradient: a=in, b=out, transpose=0, flip=1; Conv2DTranspose (kh,kw,out,in) forward: a=out, b=in,
 * transpose=0, flip=0. */
int vcg_pack_conv_kernel_bf16(const void* w, int32_t taps, int32_t a, int32_t b, int32_t transpose, int32_t flip, void* out,
                              hipStream_t stream);

/* all 3x3 64 -> 64 kernels of a model in one launch: w_host_array = HOST array of `count` (<= 48) device pointers to Keras (3,3,64,64) fp32
 * kernels; out[count][2][9][64][64] bf16: [i][0] = vcg_pack_conv_kernel_bf16(w_i, 9, 64, 64, transpose 1, flip 0) (forward),
 * [i][1] = (..., transpose 0, flip 1) (data gradient).  The pointers are read at call time (hipGraph-capturable). */
int vcg_pack_conv3x3_c64_bf16_batch(const void* const* w_host_array, int32_t count, void* out, hipStream_t stream);
/* layout + precision change at the edge of the bf16 path */
int vcg_f32_nchw_to_bf16_nhwc(const void* x, void* y, int32_t n, int32_t c, int32_t h, int32_t w, hipStream_t stream);
int vcg_bf16_nhwc_to_f32_nchw(const void* x, void* y, int32_t n, int32_t c, int32_t h, int32_t w, hipStream_t stream);

/* Conv2D forward on bf16 NHWC (replaces keras Conv2D [+BatchNormalization(inference)+PReLU+Add] at
 * upscaling/upscaler/model.py:19-25,283-285).  Instantiated: 3x3 stride 1 'same' 64->64 (generator trunk);
 * other shapes return VCG_E_UNSUPPORTED. */
int vcg_conv2d_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* w_packed, void* y, const vcg_epilogue_bf16* ep,
                        hipStream_t stream);
/* records per group the convolution writes into ep->stats for this shape and mode (> 0), or a negative VCG_E_* when its epilogue
 * cannot produce statistics for it (the caller then runs vcg_norm_stats_bf16 on the output) */
int vcg_conv2d_bf16_stats_records(const vcg_conv_desc* d, int32_t stats_mode);

/* Conv2DTranspose(strides=2, padding='same') forward on bf16 NHWC (+ fused LeakyReLU: upsampling_block,
 * upscaling/upscaler/model.py:70-75).  w_packed: [tap][out][in] (vcg_pack_conv_kernel_bf16(w, 9, out, in, 0, 0) from
 * Keras' (kh,kw,out,in)).  Instantiated: 3x3, in = 64, out = 64 * {1,2,4,8}; ep may carry shift (the bias) and VCG_ACT_LRELU. */
int vcg_conv_transpose2d_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* w_packed, void* y, const vcg_epilogue_bf16* ep,
                                  hipStream_t stream);

/* final/conv of the generator (upscaling/upscaler/model.py:290-291): Conv2D(3, 9, 'same') + bias + tanh on a bf16 NHWC
 * input with 256 channels; output fp32 NCHW (what vcg_nchw_to_frames_u8 consumes).  wfrag: the kernel re-laid out as
 * MFMA operand fragments by vcg_pack_final9x9_bf16 from Keras' (9,9,256,3) fp32 kernel: VCG_FINAL9X9_WFRAG_BYTES bytes
 * (9216 fragments of 16 bytes + 64 zero bytes the kernel fetches its padding pixels from). */
#define VCG_FINAL9X9_WFRAG_BYTES ((4 * 9 * 4 * 64 + 4) * 16)
int vcg_pack_final9x9_bf16(const void* w, void* out, hipStream_t stream);
int vcg_conv9x9_to3_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, int32_t tanh_act, void* y,
                             hipStream_t stream);

/* initial/conv of the generator (upscaling/upscaler/model.py:275-276): Conv2D(64, 9, 'same') + bias + PReLU from the
 * fp32 NCHW frames to bf16 NHWC -- the entry into the bf16 layout.  wfrag: VCG_FIRST9X9_WFRAG_BYTES bytes of MFMA operand
 * fragments made by vcg_pack_first9x9_bf16 from Keras' (9,9,3,64) fp32 kernel; prelu_alpha may be NULL (no activation). */
#define VCG_FIRST9X9_WFRAG_BYTES (27 * 64 * 2 * 16)
int vcg_pack_first9x9_bf16(const void* w, void* out, hipStream_t stream);
/* the same fragment layout for a 3 -> cout convolution, cout = 64*{1,2,4,8} (cout/64 x VCG_FIRST9X9_WFRAG_BYTES bytes):
 * dgrad = 0: w is Keras' (9,9,3,cout) kernel, packed for the forward pass; dgrad = 1: w is Keras' (9,9,cout,3) kernel of a
 * cout -> 3 convolution (final/conv, model.py:290), packed for its data gradient (taps flipped, roles of in/out swapped) */
int vcg_pack_conv9x9_3ch_bf16(const void* w, int32_t cout, int32_t dgrad, void* out, hipStream_t stream);
/* data gradient of final/conv on the bf16 path: dy fp32 NCHW [n,3,h,w] -> dx bf16 NHWC [n,h,w,cin]; d describes the FORWARD
 * convolution (cin = 256, cout = 3).  y_prev (optional): the bf16 NHWC output of the LeakyReLU feeding the convolution
 * (upsampling_block, model.py:73); dx is then multiplied by its derivative, i.e. it is the gradient in front of the activation. */
int vcg_conv9x9_to3_bf16_dgrad(const vcg_conv_desc* d, const void* dy, const void* wfrag, const void* y_prev, float lrelu_slope, void* dx,
                               hipStream_t stream);
/* the same, also leaving per-channel sums of the stored dx as fp32 records [vcg_conv9x9_to3_bf16_dgrad_chsum_records(d)][cin]: added up by
 * vcg_sum_records they are the bias gradient of the layer that produced the convolution's input (upsampling_block's Conv2DTranspose,
 * model.py:72) -- no separate pass over dx */
int vcg_conv9x9_to3_bf16_dgrad_chsum_records(const vcg_conv_desc* d);
int vcg_conv9x9_to3_bf16_dgrad_chsum(const vcg_conv_desc* d, const void* dy, const void* wfrag, const void* y_prev, float lrelu_slope, void* dx,
                                     float* records, hipStream_t stream);
int vcg_conv9x9_from3_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, const void* prelu_alpha,
                               void* y, hipStream_t stream);

/* Conv2D on THREE input channels from the fp32 NCHW frames to bf16 NHWC, + bias + LeakyReLU(lrelu_slope; 1 = none): the critics' first
 * layer where it enters the bf16 layout -- Conv2D(64, 3) 'same' (upscaling/upscaler/model.py:839, simple_512 / thin_512; its BatchNormalization
 * follows on bf16) and the PatchGAN's Conv2D(64, 4, strides 2) + LeakyReLU(0.2).  3x3 stride 1 and 4x4 stride 2, cout = 64*{1,2,4,8}; d carries
 * the pads.  wfrag: vcg_conv3ch_bf16_wfrag_bytes(kh, kw, cout) bytes made by vcg_pack_conv3ch_bf16 from Keras' (kh,kw,3,cout) kernel. */
size_t vcg_conv3ch_bf16_wfrag_bytes(int32_t kh, int32_t kw, int32_t cout);
int vcg_pack_conv3ch_bf16(const void* w, int32_t kh, int32_t kw, int32_t cout, void* out, hipStream_t stream);
int vcg_conv3ch_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, float lrelu_slope, void* y, hipStream_t stream);
/* data gradient of such a first layer from the bf16 NHWC gradient dz [n][oh][ow][64] in front of its activation to the fp32 NCHW gradient
 * of the frames [n][3][h][w] (what the generator's backward consumes): the 3-channel result is computed as 12 (4x4 stride 2: four sub-pixel
 * phases x 3) or 3 of 64 virtual output channels of a 3x3 stride-1 convolution over dz on vcg_conv2d_nhwc_bf16_fwd and scattered.
 * cout = 64, pads 1; w_hwio: Keras' (kh,kw,3,64) kernel (fp32; its bf16 copy is the operand). */
size_t vcg_conv3ch_bf16_dgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv3ch_bf16_dgrad(const vcg_conv_desc* d, const void* dz, const float* w_hwio, float* dx, void* ws, size_t ws_bytes, vcg_stream_t stream);
/* weight (+ bias) gradient of a layer that reads the 3-channel frames -- initial/conv (9x9, upscaling/upscaler/model.py:275), the critics'
 * first layers (3x3 'same', model.py:839, 904; the PatchGAN's 4x4 stride 2) -- from the fp32 NCHW frames x [n][3][h][w] and the bf16 NHWC gradient
 * dz [n][oh][ow][cout] in front of the layer's activation: both as bf16 MFMA operands (the roundings the forward kernel multiplies), fp32
 * accumulation, dw in Keras' (kh,kw,3,cout) layout, dbias [cout] or NULL; deterministic.  cout % 64 == 0, even width; VCG_E_UNSUPPORTED
 * otherwise (run vcg_conv2d_wgrad on an fp32 copy of dz).  Replaces the gradient of model.py:275 / :839 that Keras derives. */
size_t vcg_conv3ch_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv3ch_bf16_wgrad(const vcg_conv_desc* d, const float* x, const void* dz, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                           vcg_stream_t stream);
/* the training-mode form of vcg_conv9x9_from3_bf16_fwd: also stores z, the value in front of the PReLU (bf16 NHWC like y), which
 * vcg_prelu_bwd_nhwc_bf16 needs */
int vcg_conv9x9_from3_bf16_fwd_train(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, const void* prelu_alpha,
                                     void* y, void* z, hipStream_t stream);
/* backward of that PReLU (initial/prelu, upscaling/upscaler/model.py:276) where the bf16 trunk begins: d1 (+ d2, optional: the long skip's
 * gradient, model.py:285) bf16 NHWC [n][hw][c] -> dz_nchw fp32 [n][c][hw] = (d1 + d2) * (z >= 0 ? 1 : alpha[c]) -- the layout
 * vcg_conv2d_wgrad reads -- and records [vcg_prelu_bwd_nhwc_bf16_records(n, hw)][c] whose sum (vcg_sum_records) is the slope gradient
 * sum((d1 + d2) * z, z < 0).  c % 8 == 0. */
int vcg_prelu_bwd_nhwc_bf16_records(int n, int hw);
int vcg_prelu_bwd_nhwc_bf16(const void* d1, const void* d2, const void* z, const float* prelu_alpha, int n, int c, int hw, float* dz_nchw,
                            float* records, hipStream_t stream);
/* the same with dz left as bf16 NHWC [n][hw][c] (what vcg_conv3ch_bf16_wgrad reads); same records */
int vcg_prelu_bwd_nhwc_bf16_to_bf16(const void* d1, const void* d2, const void* z, const float* prelu_alpha, int n, int c, int hw, void* dz_nhwc,
                                    float* records, hipStream_t stream);

/* BatchNormalization / instance norm (+PReLU / LeakyReLU, +Add) on bf16 NHWC activations (model.py:20-25): statistics
 * and arithmetic in fp32.  vcg_norm_stats_bf16: per-channel (VCG_NORM_BATCH) or per-(n,c) (VCG_NORM_INSTANCE) mean and
 * biased variance, c % 8 == 0, c <= 256; feed them to vcg_norm_finalize (shared with the fp32 path) for scale / shift and
 * the moving averages, then vcg_norm_act_fwd_bf16: y = act(x*scale + shift) + residual, scale/shift [c] or [n*c]. */
size_t vcg_norm_stats_bf16_workspace_bytes(int n, int c, int hw, int mode);
int vcg_norm_stats_bf16(const void* x, int n, int c, int hw, int mode, float* mean, float* var, void* ws, size_t ws_bytes,
                        hipStream_t stream);
int vcg_norm_act_fwd_bf16(const void* x, int n, int c, int hw, const float* scale, const float* shift, int per_sample, int act,
                          float alpha, const float* prelu_alpha, const void* residual, void* y, hipStream_t stream);
/* backward, same contract as vcg_norm_act_bwd (x = the layer's INPUT, statistics as saved by the forward pass; dx bf16 NHWC;
 * dgamma / dbeta / dprelu_alpha fp32 [c] or NULL) */
size_t vcg_norm_act_bwd_bf16_workspace_bytes(int n, int c, int hw, int mode);
int vcg_norm_act_bwd_bf16(const void* x, const void* dy, int n, int c, int hw, int mode, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int act, float act_alpha, const float* prelu_alpha, int use_batch_stats,
                          void* dx, float* dgamma, float* dbeta, float* dprelu_alpha, void* ws, size_t ws_bytes, hipStream_t stream);

/* weight (and bias) gradient of the bf16 3x3 stride-1 'same' 64->64 convolution: x, dy bf16 NHWC; dw fp32 in Keras' (3,3,in,out)
 * layout (it accumulates into the fp32 master-weight gradient like vcg_conv2d_wgrad), dbias [64] fp32 or NULL. */
/* ---- one-output-channel Conv2D on bf16 NHWC: the 70x70 PatchGAN's last layer, Conv2D(1, 4) on 512 channels (north_star extension, SURVEY.md
 *      section 8 row a11; follows the reference's critic factories, upscaling/upscaler/model.py:836-896).  x / dx: bf16 [n][h][w][cin]
 *      (cin a multiple of 8, <= 512), y / dy: fp32 [n][1][oh][ow], w_hwio: the Keras kernel (kh,kw,cin,1) in fp32; 3x3 / 4x4, stride 1,
 *      any zero padding.  VCG_E_UNSUPPORTED otherwise (the caller then uses vcg_conv2d_* on fp32 copies). ------------------------------ */
int vcg_conv2d_cout1_nhwc_bf16_fwd(const vcg_conv_desc* d, const void* x, const float* w_hwio, const float* bias, float* y, vcg_stream_t stream);
int vcg_conv2d_cout1_nhwc_bf16_dgrad(const vcg_conv_desc* d, const float* dy, const float* w_hwio, void* dx, vcg_stream_t stream);
size_t vcg_conv2d_cout1_nhwc_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d);
/* dw_hwio and dbias (may be NULL) are overwritten; deterministic (fixed-order sum of per-workgroup records) */
int vcg_conv2d_cout1_nhwc_bf16_wgrad(const vcg_conv_desc* d, const void* x, const float* dy, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                                     vcg_stream_t stream);

size_t vcg_conv2d_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv2d_bf16_wgrad(const vcg_conv_desc* d, const void* x, const void* dy, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                          hipStream_t stream);

/* weight gradient of final/conv -- Conv2D(3, 9, 'same') on 256 channels (upscaling/upscaler/model.py:290) -- in the bf16 configs:
 * x bf16 NHWC [n][h][w][256], dz fp32 NCHW [n][3][h][w] (the gradient behind the tanh; rounded to bf16 as an MFMA operand), dw fp32 in
 * Keras' (9,9,256,3) layout, overwritten.  w must be even.  (The bias gradient is the channel sum of dz: vcg_act_bwd / vcg_channel_sum.) */
size_t vcg_conv9x9_to3_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv9x9_to3_bf16_wgrad(const vcg_conv_desc* d, const void* x, const float* dz, float* dw_hwio, void* ws, size_t ws_bytes, hipStream_t stream);

/* ---- generic bf16 NHWC Conv2D (any 3x3 / 4x4 / 5x5, stride 1-3, channels multiples of 32 / 16): the discriminators' layers in
 * the bf16 configs (upscaling/upscaler/model.py:839-871, 904-936; PatchGAN) and the generator's Conv2DTranspose gradients.
 * Weights as MFMA operand fragments: vcg_pack_conv_frag_bf16(w, taps, mdim, kdim, mode, out), out = taps*mdim*kdim bf16:
 *   mode 0: w is a Keras (kh,kw,in,out) kernel, packed for the forward pass       (mdim = out, kdim = in)
 *   mode 1: the same kernel packed for its data gradient                           (mdim = in,  kdim = out)
 * (a Conv2DTranspose kernel (kh,kw,out,in) is a Conv2D kernel with the roles of in / out exchanged.) */
size_t vcg_conv_frag_bf16_bytes(int taps, int mdim, int kdim);
int vcg_pack_conv_frag_bf16(const float* w, int taps, int mdim, int kdim, int mode, void* out, hipStream_t stream);
/* both copies of one Keras (kh,kw,in,out) kernel in one launch: out_fwd = mode 0 with (mdim, kdim) = (cout, cin), out_dgrad = mode 1 with
 * (mdim, kdim) = (cin, cout); cin, cout multiples of 32 */
int vcg_pack_conv_frag_bf16_pair(const float* w, int32_t taps, int32_t cin, int32_t cout, void* out_fwd, void* out_dgrad, hipStream_t stream);
/* y = act(conv(x) + bias), x / y bf16 NHWC, bias fp32 [cout] or NULL, act VCG_ACT_NONE / VCG_ACT_LRELU */
int vcg_conv2d_nhwc_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const float* bias, int act, float act_alpha,
                             void* y, hipStream_t stream);
/* the same without activation, also leaving per-tile statistics of the stored output for the normalisation behind the layer
 * (model.py:840-841 and the PatchGAN blocks in training mode): stats fp32 [n][records][2][cout], records per image =
 * vcg_conv2d_nhwc_bf16_stats_records(d, VCG_STATS_INSTANCE) (one per 8x32-pixel output tile; ..._BATCH returns n times that) -- read by
 * vcg_norm_finalize_partials as [1][n*records] (batch statistics) or [n][records] (instance norm).  A negative record count means the
 * tiled kernel does not serve the shape: use vcg_conv2d_nhwc_bf16_fwd + vcg_norm_stats_bf16. */
int vcg_conv2d_nhwc_bf16_stats_records(const vcg_conv_desc* d, int stats_mode);
int vcg_conv2d_nhwc_bf16_fwd_stats(const vcg_conv_desc* d, const void* x, const void* wfrag, const float* bias, void* y, float* stats,
                                   hipStream_t stream);
/* Conv2DTranspose(k, strides 2, 'same') + bias + LeakyReLU (model.py:72-73) on bf16 NHWC as the data gradient of the stride-2 convolution its
 * kernel is: d describes the transposed layer (cin, h, w -> cout, oh, ow; pads = the 'same' crop), wfrag =
 * vcg_pack_conv_frag_bf16(kernel (kh,kw,out,in), k*k, mdim = cout, kdim = cin, mode 1) */
int vcg_conv_transpose2d_nhwc_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const float* bias, int act, float act_alpha,
                                       void* y, hipStream_t stream);
/* dx = data gradient of the layer d describes (d is the FORWARD layer), dy / dx bf16 NHWC, wfrag_t packed with mode 1.
 * mask_src (optional, bf16 NHWC of dx's shape): dx *= (mask_src > 0 ? 1 : mask_slope), the derivative of a LeakyReLU whose
 * OUTPUT mask_src is and which feeds the layer. */
int vcg_conv2d_nhwc_bf16_dgrad(const vcg_conv_desc* d, const void* dy, const void* wfrag_t, const void* mask_src, float mask_slope,
                               void* dx, hipStream_t stream);
/* weight / bias gradient, x [n][h][w][cin] and dy [n][oh][ow][cout] bf16 NHWC, 3x3 / 4x4, stride 1 / 2, channels multiples of 64;
 * dw fp32 in Keras' (kh,kw,in,out) layout, dbias fp32 [cout] or NULL (deterministic: fixed-order sum of per-workgroup partials) */
size_t vcg_conv2d_nhwc_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv2d_nhwc_bf16_wgrad(const vcg_conv_desc* d, const void* x, const void* dy, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                               hipStream_t stream);
/* Conv2DTranspose(strides=2,'same') (upscaling/upscaler/model.py:72): weight gradient from the layer input x [n][h][w][cin] and the
 * gradient dz [n][oh][ow][cout] in front of its activation, both bf16 NHWC; dw fp32 in Keras' (kh,kw,out,in) layout */
size_t vcg_conv_transpose2d_nhwc_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d);
int vcg_conv_transpose2d_nhwc_bf16_wgrad(const vcg_conv_desc* d, const void* x, const void* dz, float* dw_hwoi, void* ws, size_t ws_bytes,
                                         hipStream_t stream);
/* flat precision changes (Flatten of an NHWC tensor IS its memory order: the Dense head of the discriminators stays fp32) */
int vcg_bf16_to_f32(const void* x, float* y, size_t count, hipStream_t stream);
int vcg_f32_to_bf16(const float* x, void* y, size_t count, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VCG_H */
