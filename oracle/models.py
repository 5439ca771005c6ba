"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/keras_ops.py header).

CPU restatement of the reference's generator / discriminator topologies:
  make_upscaler_orig              upscaling/upscaler/model.py:267-295  (blocks :15-27, :70-75)
  make_discriminator_simple_512   upscaling/upscaler/model.py:836-896
  make_discriminator_thin_512     upscaling/upscaler/model.py:901-961
plus the north_star extension with no reference counterpart:
  make_discriminator_patchgan_70  (C64-C128-C256 k4 s2, C512 k4 s1, C1 k4 s1, pad 1, LReLU 0.2)

Weights live in a flat dict  "<keras layer name>/<keras weight name>" -> torch tensor, in Keras'
own layouts, so the dict doubles as the weight-exchange format with the product.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import keras_ops as K


# ----------------------------------------------------------------------------------------------
# weight construction (Keras defaults: Glorot-uniform kernels, zero bias, BN gamma=1 beta=0
# moving_mean=0 moving_variance=1, PReLU alpha=0)
# ----------------------------------------------------------------------------------------------
def _conv_w(w, rng, name, kh, kw, cin, cout):
    w[name + "/kernel"] = K.glorot_uniform(rng, (kh, kw, cin, cout), kh * kw * cin, kh * kw * cout)
    w[name + "/bias"] = np.zeros((cout,), np.float32)


def _convt_w(w, rng, name, kh, kw, cin, cout):
    # Keras Conv2DTranspose kernel is (kh, kw, out, in); fan_in/fan_out as keras computes them for
    # that shape: receptive*shape[-2] / receptive*shape[-1]
    w[name + "/kernel"] = K.glorot_uniform(rng, (kh, kw, cout, cin), kh * kw * cout, kh * kw * cin)
    w[name + "/bias"] = np.zeros((cout,), np.float32)


def _dense_w(w, rng, name, cin, cout):
    w[name + "/kernel"] = K.glorot_uniform(rng, (cin, cout), cin, cout)
    w[name + "/bias"] = np.zeros((cout,), np.float32)


def _bn_w(w, name, c):
    w[name + "/gamma"] = np.ones((c,), np.float32)
    w[name + "/beta"] = np.zeros((c,), np.float32)
    w[name + "/moving_mean"] = np.zeros((c,), np.float32)
    w[name + "/moving_variance"] = np.ones((c,), np.float32)


def _prelu_w(w, name, c):
    w[name + "/alpha"] = np.zeros((c,), np.float32)


NON_TRAINABLE_SUFFIXES = ("/moving_mean", "/moving_variance")


def is_trainable(name):
    return not name.endswith(NON_TRAINABLE_SUFFIXES)


def to_torch(w, dtype=torch.float32, requires_grad=False):
    out = OrderedDict()
    for k, v in w.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and is_trainable(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# ----------------------------------------------------------------------------------------------
# generator: make_upscaler_orig (model.py:267-295)
# ----------------------------------------------------------------------------------------------
def init_upscaler_orig(output_image_shape, kernel_size=5, filters=64, upscale_factor=4,
                       res_block_num=16, seed=7, norm="batch"):
    rng = np.random.RandomState(seed)
    k = kernel_size
    w = OrderedDict()
    _conv_w(w, rng, "initial/conv", 9, 9, output_image_shape[2], filters)
    _prelu_w(w, "initial/prelu", filters)
    for i in range(res_block_num):
        n = "res_block/%d" % i
        _conv_w(w, rng, n + "/conv_pre", k, k, filters, filters)
        if norm == "batch":
            _bn_w(w, n + "/batch_norm_pre", filters)
        _prelu_w(w, n + "/prelu", filters)
        _conv_w(w, rng, n + "/conv_post", k, k, filters, filters)
        if norm == "batch":
            _bn_w(w, n + "/batch_norm_post", filters)
    _conv_w(w, rng, "prefinal/conv2d", k, k, filters, 64)         # 64 hard-coded: model.py:283
    if norm == "batch":
        _bn_w(w, "prefinal/batch_norm", 64)
    cin = 64
    for i in range(int(math.log(upscale_factor, 2))):
        _convt_w(w, rng, "upscaling/%d/block/conv_transp" % i, k, k, cin, 256)  # 256: model.py:288
        cin = 256
    _conv_w(w, rng, "final/conv", 9, 9, cin, 3)
    return w


def upscaler_orig_forward(w, x_nhwc, training, res_block_num, upscale_factor, taps=None, norm="batch", trunk_bf16=False, tail_bf16=False,
                          fold_inference=True, init_bf16=None):
    """x_nhwc: [N,h,w,3] -> ([N,h*f,w*f,3], bn_updates).  ``training`` selects batch vs moving BN
    statistics (Keras learning phase: predict=0, train_on_batch=1).  ``taps`` (optional dict)
    receives named NCHW intermediates for kernel-level parity tests.  ``trunk_bf16`` marks the tensors the
    product's ``trunk_dtype='bf16'`` mode stores in bf16 (values and gradients; keras_ops.bf16_*); ``tail_bf16`` adds those of
    ``'bf16+tail'`` (up-sampling block and final/conv).  ``fold_inference``: in learning phase 0 the product's bf16 trunk applies the
    BatchNormalization in the convolution's epilogue, so the convolution's own output is never stored (no rounding there).
    ``init_bf16`` (default: as ``tail_bf16``): initial/conv reads bf16 copies of the frames and of its kernel (the product's
    InitialConv9x9Bf16; fp32 accumulation, fp32 weight gradient from the fp32 frames -- the straight-through gradient of the rounding)."""
    if init_bf16 is None:
        init_bf16 = tail_bf16
    upd = OrderedDict()
    st = K.bf16_store if trunk_bf16 else (lambda v: v)
    rf = K.bf16_round_fwd if trunk_bf16 else (lambda v: v)
    rg = K.bf16_round_grad if trunk_bf16 else (lambda v: v)

    def tconv(x, name):          # trunk convolution: bf16 copy of the fp32 master kernel, output (+bias) stored in bf16
        y = K.conv2d(x, rf(w[name + "/kernel"]), w[name + "/bias"], 1, "same")
        return y if (fold_inference and not training and norm == "batch") else st(y)

    def bn(x, name):
        if norm == "instance":          # north_star extension (no reference counterpart)
            return K.instancenorm(x)
        y, mm, mv = K.batchnorm(x, w[name + "/gamma"], w[name + "/beta"], w[name + "/moving_mean"],
                                w[name + "/moving_variance"], training)
        if training:
            upd[name + "/moving_mean"], upd[name + "/moving_variance"] = mm, mv
        return y

    def conv(x, name, stride=1):
        return K.conv2d(x, w[name + "/kernel"], w[name + "/bias"], stride, "same")

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = x_nhwc.permute(0, 3, 1, 2)
    if init_bf16:
        c0 = K.conv2d(K.bf16_round_fwd(x), K.bf16_round_fwd(w["initial/conv/kernel"]), w["initial/conv/bias"], 1, "same")
    else:
        c0 = conv(x, "initial/conv")
    m = tap("initial/prelu", K.prelu(tap("initial/conv", c0), w["initial/prelu/alpha"]))
    skip = rf(m)                     # long skip: bf16 copy of the fp32 tensor, its gradient stays fp32
    m = st(m)
    for i in range(res_block_num):
        n = "res_block/%d" % i
        gen = m
        m = tap(n + "/conv_pre", tconv(m, n + "/conv_pre"))
        m = bn(m, n + "/batch_norm_pre")
        m = tap(n + "/prelu", st(K.prelu(m, w[n + "/prelu/alpha"])))
        m = tap(n + "/conv_post", tconv(m, n + "/conv_post"))
        m = bn(m, n + "/batch_norm_post")
        m = tap(n + "/final_add", st(gen + m))
    m = tconv(m, "prefinal/conv2d")
    m = rg(bn(m, "prefinal/batch_norm"))
    # Add misnamed in model.py:285.  'bf16+tail': the up-sampling block's data gradient is produced in bf16, so BOTH branches of
    # the add (the long skip too) see the rounded gradient
    m = tap("prefinal/tanh", (K.bf16_store if tail_bf16 else rf)(skip + m))
    trf = K.bf16_round_fwd if tail_bf16 else (lambda v: v)
    trg = K.bf16_round_grad if tail_bf16 else (lambda v: v)
    for i in range(int(math.log(upscale_factor, 2))):
        n = "upscaling/%d/block" % i
        m = K.conv2d_transpose_same(m, trf(w[n + "/conv_transp/kernel"]), w[n + "/conv_transp/bias"], 2)
        m = tap(n + "/leaky_relu", trf(K.leaky_relu(trg(m), 0.2)))      # output stored in bf16; the gradient IN FRONT of the
        #                                                                 activation is what the product stores in bf16
    m = tap("final/conv", K.conv2d(m, trf(w["final/conv/kernel"]), w["final/conv/bias"], 1, "same"))
    m = torch.tanh(m)
    return m.permute(0, 2, 3, 1), upd


# ----------------------------------------------------------------------------------------------
# generator: make_upscaler_attention (model.py:299-328; blocks :30-48, :78-98) -- train_gan3.py's default ('resnet-att', :55)
# ----------------------------------------------------------------------------------------------
def init_upscaler_attention(output_image_shape, kernel_size=5, filters=64, upscale_factor=4, res_block_num=16, seed=7):
    rng = np.random.RandomState(seed)
    k, c = kernel_size, output_image_shape[2]
    w = OrderedDict()
    _conv_w(w, rng, "initial/conv", 9, 9, c, filters)
    _prelu_w(w, "initial/prelu", filters)
    for i in range(res_block_num):
        n = "res_block/%d" % i
        _conv_w(w, rng, n + "/attention", k, k, c, filters)
        _conv_w(w, rng, n + "/conv_pre", k, k, filters, filters)
        _bn_w(w, n + "/batch_norm_pre", filters)
        _prelu_w(w, n + "/prelu", filters)
        _conv_w(w, rng, n + "/conv_post", k, k, filters, filters)
        _bn_w(w, n + "/batch_norm_post", filters)
    _conv_w(w, rng, "after_res/conv", k, k, filters, filters)
    _bn_w(w, "after_res/batch_norm", filters)
    cin = filters
    for i in range(int(math.log(upscale_factor, 2))):
        n, scale = "upscaling/%d/block" % i, 2 ** (i + 1)
        _conv_w(w, rng, n + "/attention", k, k, 2 * c, cin)                                  # on [nearest, bilinear] of the input
        _convt_w(w, rng, n + "/conv_transp", k, k, cin, 128)
        _convt_w(w, rng, n + "/to_add_input_conv_transp", scale + 1, scale + 1, c, 128)
        cin = 128
    _conv_w(w, rng, "final/conv", 9, 9, cin, 3)
    return w


def resize_images_tf1(x, factor, interpolation):
    """K.resize_images(x, f, f, 'channels_last', interpolation) of Keras 2.2.x on TF 1.14: tf.image.resize_nearest_neighbor /
    resize_bilinear with align_corners=False and NO half-pixel centres (source coordinate = destination * in/out)."""
    if factor == 1:
        return x
    n, c, h, w = x.shape
    oy = torch.arange(h * factor, dtype=x.dtype) / factor
    ox = torch.arange(w * factor, dtype=x.dtype) / factor
    if interpolation == "nearest":
        return x[:, :, oy.floor().long()][:, :, :, ox.floor().long()]
    y0, x0 = oy.floor().long(), ox.floor().long()
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    fy, fx = (oy - y0).view(1, 1, -1, 1), (ox - x0).view(1, 1, 1, -1)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy) + bot * fy


def upscaler_attention_forward(w, x_nhwc, training, res_block_num, upscale_factor, taps=None):
    """[N,h,w,3] -> ([N,h*f,w*f,3], bn_updates)"""
    upd = OrderedDict()

    def bn(x, name):
        y, mm, mv = K.batchnorm(x, w[name + "/gamma"], w[name + "/beta"], w[name + "/moving_mean"], w[name + "/moving_variance"], training)
        if training:
            upd[name + "/moving_mean"], upd[name + "/moving_variance"] = mm, mv
        return y

    def conv(x, name, stride=1):
        return K.conv2d(x, w[name + "/kernel"], w[name + "/bias"], stride, "same")

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = x_nhwc.permute(0, 3, 1, 2)
    m = K.prelu(conv(x, "initial/conv"), w["initial/prelu/alpha"])
    skip = m
    for i in range(res_block_num):
        n = "res_block/%d" % i
        gen = m
        m = torch.sigmoid(conv(x, n + "/attention")) * m                                      # model.py:34-36
        m = K.prelu(bn(conv(m, n + "/conv_pre"), n + "/batch_norm_pre"), w[n + "/prelu/alpha"])
        m = bn(conv(m, n + "/conv_post"), n + "/batch_norm_post")
        m = tap(n + "/final_add", gen + m)
    m = tap("after_res/add", skip + bn(conv(m, "after_res/conv"), "after_res/batch_norm"))
    for i in range(int(math.log(upscale_factor, 2))):
        n, scale = "upscaling/%d/block" % i, 2 ** (i + 1)
        up = torch.cat([resize_images_tf1(x, scale // 2, "nearest"), resize_images_tf1(x, scale // 2, "bilinear")], 1)   # :80-82
        m = torch.sigmoid(conv(up, n + "/attention")) * m                                                                  # :86-89
        m = K.leaky_relu(K.conv2d_transpose_same(m, w[n + "/conv_transp/kernel"], w[n + "/conv_transp/bias"], 2), 0.2)
        t = torch.atanh(0.99999 * x)                                                                                       # :94
        t = K.conv2d_transpose_same(t, w[n + "/to_add_input_conv_transp/kernel"], w[n + "/to_add_input_conv_transp/bias"], scale)
        m = tap(n + "/add_input", m + t)
    m = torch.tanh(conv(m, "final/conv"))
    return m.permute(0, 2, 3, 1), upd


# ----------------------------------------------------------------------------------------------
# discriminators
# ----------------------------------------------------------------------------------------------
_SIMPLE_FILTERS = (64, 128, 256, 512, 512, 512, 512, 512, 512)     # model.py:839-871
_THIN_FILTERS = (64,) + (128,) * 8                                      # model.py:904-936


def _disc_out_hw(h, w, nblocks):
    for i in range(1, nblocks):
        h, w = -(-h // 2), -(-w // 2)
    return h, w


def init_discriminator_512(input_shape, variant="simple", seed=11):
    filters = _SIMPLE_FILTERS if variant == "simple" else _THIN_FILTERS
    rng = np.random.RandomState(seed)
    w = OrderedDict()
    cin = input_shape[2]
    for i, f in enumerate(filters):
        n = "discriminator/block_%d" % (i + 1)
        _conv_w(w, rng, n + "/Conv2d", 3, 3, cin, f)
        _bn_w(w, n + "/BatchNorm", f)
        cin = f
    h, ww = _disc_out_hw(input_shape[0], input_shape[1], len(filters))
    _dense_w(w, rng, "discriminator/final/Dense_1", h * ww * cin, 1024)
    _bn_w(w, "discriminator/final/BatchNorm_1", 1024)
    _dense_w(w, rng, "discriminator/final/Dense_2", 1024, 32)
    _bn_w(w, "discriminator/final/BatchNorm_2", 32)
    _dense_w(w, rng, "discriminator/final/Dense_3", 32, 1)
    return w


def discriminator_512_forward(w, x_nhwc, training, activation="none", taps=None, bf16=False):
    """make_discriminator_simple_512 / _thin_512 forward: [N,H,W,3] -> ([N,1], bn_updates).  ``bf16`` marks the tensors the
    product's ``dtype='bf16'`` mode stores in bf16 (every block's convolution and normalisation outputs, values and gradients; bf16
    copies of the fp32 master kernels and, at block 1, of the frames) -- keras_ops.bf16_*."""
    upd = OrderedDict()
    st = K.bf16_store if bf16 else (lambda v: v)
    rf = K.bf16_round_fwd if bf16 else (lambda v: v)

    def bn(x, name):
        y, mm, mv = K.batchnorm(x, w[name + "/gamma"], w[name + "/beta"], w[name + "/moving_mean"],
                                w[name + "/moving_variance"], training)
        if training:
            upd[name + "/moving_mean"], upd[name + "/moving_variance"] = mm, mv
        return y

    m = x_nhwc.permute(0, 3, 1, 2)
    i = 1
    while ("discriminator/block_%d/Conv2d/kernel" % i) in w:
        n = "discriminator/block_%d" % i
        if i == 1:
            # block 1 enters the bf16 layout: bf16 copies of the frames and of the kernel as operands (fp32 accumulation), output stored in bf16
            m = st(K.conv2d(rf(m), rf(w[n + "/Conv2d/kernel"]), w[n + "/Conv2d/bias"], 1, "same"))
            m = st(K.leaky_relu(bn(m, n + "/BatchNorm"), 0.1))
        else:
            m = st(K.conv2d(m, rf(w[n + "/Conv2d/kernel"]), w[n + "/Conv2d/bias"], 2, "same"))
            m = st(K.leaky_relu(bn(m, n + "/BatchNorm"), 0.1))
        if taps is not None:
            taps[n + "/LeakyReLU"] = m
        i += 1
    m = K.flatten_nhwc(m)
    for j in (1, 2):
        m = K.dense(m, w["discriminator/final/Dense_%d/kernel" % j], w["discriminator/final/Dense_%d/bias" % j])
        m = bn(m, "discriminator/final/BatchNorm_%d" % j)
        m = K.leaky_relu(m, 0.1)
    m = K.dense(m, w["discriminator/final/Dense_3/kernel"], w["discriminator/final/Dense_3/bias"])
    return K.head_activation(m, activation), upd


_SPARSE_FILTERS = (64, 128, 256, 256, 256, 256)                       # model.py:967-987: 5x5 'valid', strides 1,3,3,3,3,3


def init_discriminator_sparse_512(input_shape, seed=11):
    """make_discriminator_sparse_512 (model.py:964-1012)"""
    rng = np.random.RandomState(seed)
    w = OrderedDict()
    cin, h, ww = input_shape[2], input_shape[0], input_shape[1]
    for i, f in enumerate(_SPARSE_FILTERS):
        n = "discriminator/block_%d" % (i + 1)
        _conv_w(w, rng, n + "/Conv2d", 5, 5, cin, f)
        _bn_w(w, n + "/BatchNorm", f)
        s = 1 if i == 0 else 3
        h, ww = (h - 5) // s + 1, (ww - 5) // s + 1
        cin = f
    _dense_w(w, rng, "discriminator/final/Dense_1", h * ww * cin, 128)
    _bn_w(w, "discriminator/final/BatchNorm_1", 128)
    _dense_w(w, rng, "discriminator/final/Dense_2", 128, 32)
    _bn_w(w, "discriminator/final/BatchNorm_2", 32)
    _dense_w(w, rng, "discriminator/final/Dense_3", 32, 1)
    return w


def discriminator_sparse_512_forward(w, x_nhwc, training, activation="none"):
    upd = OrderedDict()

    def bn(x, name):
        y, mm, mv = K.batchnorm(x, w[name + "/gamma"], w[name + "/beta"], w[name + "/moving_mean"], w[name + "/moving_variance"], training)
        if training:
            upd[name + "/moving_mean"], upd[name + "/moving_variance"] = mm, mv
        return y
    m = x_nhwc.permute(0, 3, 1, 2)
    for i in range(len(_SPARSE_FILTERS)):
        n = "discriminator/block_%d" % (i + 1)
        m = K.conv2d(m, w[n + "/Conv2d/kernel"], w[n + "/Conv2d/bias"], 1 if i == 0 else 3, "valid")
        m = K.leaky_relu(bn(m, n + "/BatchNorm"), 0.1)
    m = K.flatten_nhwc(m)
    for j in (1, 2):
        m = K.dense(m, w["discriminator/final/Dense_%d/kernel" % j], w["discriminator/final/Dense_%d/bias" % j])
        m = K.leaky_relu(bn(m, "discriminator/final/BatchNorm_%d" % j), 0.1)
    m = K.dense(m, w["discriminator/final/Dense_3/kernel"], w["discriminator/final/Dense_3/bias"])
    return K.head_activation(m, activation), upd


_PATCH_SPEC = ((64, 2, None), (128, 2, "norm"), (256, 2, "norm"), (512, 1, "norm"), (1, 1, "last"))


def init_discriminator_patchgan_70(input_shape, norm="instance", seed=11):
    rng = np.random.RandomState(seed)
    w = OrderedDict()
    cin = input_shape[2]
    for i, (f, s, kind) in enumerate(_PATCH_SPEC):
        n = "discriminator/block_%d" % (i + 1)
        _conv_w(w, rng, n + "/Conv2d", 4, 4, cin, f)
        if kind == "norm" and norm == "batch":
            _bn_w(w, n + "/BatchNorm", f)
        cin = f
    return w


def discriminator_patchgan_70_forward(w, x_nhwc, training, activation="none", norm="instance", taps=None, bf16=False):
    """70x70 PatchGAN: [N,H,W,3] -> ([N,H',W',1], bn_updates); 512x512 -> 62x62.  ``bf16``: storage roundings of the product's
    ``dtype='bf16'`` mode (the three normalised blocks)."""
    upd = OrderedDict()
    st = K.bf16_store if bf16 else (lambda v: v)
    rf = K.bf16_round_fwd if bf16 else (lambda v: v)
    m = x_nhwc.permute(0, 3, 1, 2)
    for i, (f, s, kind) in enumerate(_PATCH_SPEC):
        n = "discriminator/block_%d" % (i + 1)
        if kind == "norm":
            m = st(K.conv2d(m, rf(w[n + "/Conv2d/kernel"]), w[n + "/Conv2d/bias"], s, 1))
        elif i == 0:      # block 1 enters the bf16 layout: bf16 copies of the frames and of the kernel as operands, fp32 accumulation
            m = K.conv2d(rf(m), rf(w[n + "/Conv2d/kernel"]), w[n + "/Conv2d/bias"], s, 1)
        else:
            m = K.conv2d(m, w[n + "/Conv2d/kernel"], w[n + "/Conv2d/bias"], s, 1)
        if kind == "norm":
            if norm == "instance":
                m = K.instancenorm(m)
            else:
                bnn = n + "/BatchNorm"
                m, mm, mv = K.batchnorm(m, w[bnn + "/gamma"], w[bnn + "/beta"], w[bnn + "/moving_mean"],
                                        w[bnn + "/moving_variance"], training)
                if training:
                    upd[bnn + "/moving_mean"], upd[bnn + "/moving_variance"] = mm, mv
        if kind != "last":
            m = K.leaky_relu(m, 0.2)
            if kind == "norm" or i == 0:
                m = st(m)                 # block 1's output enters the bf16 layout; the normalised blocks' outputs live in it
            if taps is not None:
                taps[n + "/LeakyReLU"] = m
    m = K.head_activation(m, activation)
    return m.permute(0, 2, 3, 1), upd


def count_params(w):
    return int(sum(int(np.prod(v.shape)) for v in w.values()))


# ---------------------------------------------------------------------------------------------------------------
# keras.applications.VGG19(include_top=False) up to block5_conv4: the feature extractor of VGG_LOSS / VGG_MSE_LOSS /
# VGG_MAE_LOSS (reference: upscaling/upscaler/model.py:101-157).  The reference feeds it the [-1,1] frames as they
# are (no ImageNet preprocessing) and takes the ReLU'd output of block5_conv4.  The ImageNet weights cannot be
# fetched offline; parity runs use seeded random weights of the same shapes (He-uniform, so that activations keep
# their scale through 16 ReLU layers).
# ---------------------------------------------------------------------------------------------------------------
VGG19_BLOCKS = ((1, 2, 64), (2, 2, 128), (3, 4, 256), (4, 4, 512), (5, 4, 512))


def init_vgg19_features(seed=19, cin=3):
    rng = np.random.RandomState(seed)
    w = OrderedDict()
    for b, nconv, f in VGG19_BLOCKS:
        for i in range(nconv):
            lim = np.sqrt(6.0 / (9 * cin))
            w["block%d_conv%d/kernel" % (b, i + 1)] = rng.uniform(-lim, lim, (3, 3, cin, f)).astype(np.float32)
            w["block%d_conv%d/bias" % (b, i + 1)] = rng.uniform(-0.05, 0.05, (f,)).astype(np.float32)
            cin = f
    return w


def vgg19_block5_conv4(w, x_nhwc, taps=None):
    """VGG19 features: 3x3 'same' conv + ReLU stacks with 2x2/2 'valid' max pooling after blocks 1-4.  ``taps`` (optional list)
    receives every convolution's ReLU output (NCHW) in order."""
    x = x_nhwc.permute(0, 3, 1, 2)
    for b, nconv, _ in VGG19_BLOCKS:
        for i in range(nconv):
            n = "block%d_conv%d" % (b, i + 1)
            x = torch.relu(K.conv2d(x, w[n + "/kernel"], w[n + "/bias"], 1, "same"))
            if taps is not None:
                taps.append(x)
        if b < 5:
            x = torch.nn.functional.max_pool2d(x, 2, 2)
    return x.permute(0, 2, 3, 1)
