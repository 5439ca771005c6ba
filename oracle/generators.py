"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/keras_ops.py header).

CPU restatement of the other generator topologies train_gan3.py offers behind ``-gm`` (upscaling/train_gan3.py:55,234-252):
  make_upscaler_skip_con          upscaling/upscaler/model.py:332-363
  make_upscaler_unetish           upscaling/upscaler/model.py:570-634   (blocks :505-566)
  make_upscaler_unetish_add       upscaling/upscaler/model.py:642-716
  make_upscaler_unetish_complex   upscaling/upscaler/model.py:743-827
  make_upscaler_incep_resnet      upscaling/upscaler/model.py:443-497   (blocks :372-440)

Written define-by-run: ``Net`` looks every layer's weights up by its Keras name in a flat dict (the weight-exchange format of
oracle/models.py); run with ``rng`` set, missing weights are created the way Keras initialises them, so one function both counts /
initialises the parameters and evaluates the network.  Unnamed layers get Keras' automatic names (conv2d_1, batch_normalization_3,
p_re_lu_1 ...: keras.backend.get_uid per class, counted from 1 in a fresh session).  Dropout is restated as tf.nn.dropout GIVEN the
keep-mask (x * mask / (1 - rate)): the reference's random stream is not reproducible, the arithmetic is.
"""
import math
from collections import OrderedDict

import torch

from . import keras_ops as K
from . import models as M


class Net:
    def __init__(self, w, training, rng=None, masks=None):
        self.w, self.training, self.rng, self.masks = w, training, rng, masks or {}
        self.uid = {}
        self.names = set()
        self.upd = OrderedDict()          # moving statistics after the step (learning phase 1)
        self.dtype = torch.float64
        self.norm = "batch"               # 'instance': bn() is the non-affine instance norm of the north_star extension (no parameters)

    # -- names ------------------------------------------------------------------------------------------
    def _name(self, name, cls):
        if not name:
            self.uid[cls] = self.uid.get(cls, 0) + 1
            name = "%s_%d" % (cls, self.uid[cls])
        if name in self.names:
            # keras.engine.network.Network._init_graph_network: all layer names should be unique
            raise ValueError('The name "%s" is used more than once in the model. All layer names should be unique.' % name)
        self.names.add(name)
        return name

    def _get(self, key, make):
        if key not in self.w:
            if self.rng is None:
                raise KeyError(key)
            tmp = OrderedDict()
            make(tmp)
            for k, v in tmp.items():
                self.w[k] = torch.tensor(v, dtype=self.dtype)
        return self.w[key]

    # -- layers -------------------------------------------------------------------------------------------
    def conv2d(self, x, filters, k, stride=1, name=None):
        n = self._name(name, "conv2d")
        cin = x.shape[1]
        kh, kw = k if isinstance(k, tuple) else (k, k)
        self._get(n + "/kernel", lambda t: M._conv_w(t, self.rng, n, kh, kw, cin, filters))
        return K.conv2d(x, self.w[n + "/kernel"], self.w[n + "/bias"], stride, "same")

    def conv2d_transpose(self, x, filters, k, stride=2, name=None):
        n = self._name(name, "conv2d_transpose")
        cin = x.shape[1]
        self._get(n + "/kernel", lambda t: M._convt_w(t, self.rng, n, k, k, cin, filters))
        return K.conv2d_transpose_same(x, self.w[n + "/kernel"], self.w[n + "/bias"], stride)

    def bn(self, x, name=None):
        n = self._name(name, "batch_normalization")
        if self.norm == "instance":
            return K.instancenorm(x)
        c = x.shape[1]
        self._get(n + "/gamma", lambda t: M._bn_w(t, n, c))
        y, mm, mv = K.batchnorm(x, self.w[n + "/gamma"], self.w[n + "/beta"], self.w[n + "/moving_mean"], self.w[n + "/moving_variance"], self.training)
        if self.training:
            self.upd[n + "/moving_mean"], self.upd[n + "/moving_variance"] = mm, mv
        return y

    def prelu(self, x, name=None):
        n = self._name(name, "p_re_lu")
        c = x.shape[1]
        self._get(n + "/alpha", lambda t: M._prelu_w(t, n, c))
        if n in self.masks:          # the activation mask of another evaluation (x >= 0 there): the same branch of the function, even for
            #                          pre-activations within rounding of 0
            return torch.where(self.masks[n].bool(), x, x * self.w[n + "/alpha"].view(1, -1, 1, 1))
        return K.prelu(x, self.w[n + "/alpha"])

    def dropout(self, x, rate, name=None):
        n = self._name(name, "dropout")
        if not self.training or not 0.0 < rate < 1.0:          # keras.layers.Dropout.call
            return x
        return x * self.masks[n].to(x.dtype) / (1.0 - rate)      # tf.nn.dropout with the given keep-mask

    @staticmethod
    def crop(x, cropping):
        (t, b), (l, r) = cropping
        return x[:, :, t:x.shape[2] - b, l:x.shape[3] - r]


# ---- model.py:15-27, 70-75 -------------------------------------------------------------------------------------
def residual_block(net, model, kernel_size, filters, strides, name=""):
    gen = model
    model = net.conv2d(model, filters, kernel_size, strides, name + "/conv_pre")
    model = net.bn(model, name + "/batch_norm_pre")
    model = net.prelu(model, name + "/prelu")
    model = net.conv2d(model, filters, kernel_size, strides, name + "/conv_post")
    model = net.bn(model, name + "/batch_norm_post")
    net._name(name + "/final_add", "add")
    return gen + model


def upsampling_block(net, model, kernel_size, filters, strides, name=""):
    model = net.conv2d_transpose(model, filters, kernel_size, strides, name + "/conv_transp")
    net._name(name + "/leaky_relu", "leaky_re_lu")
    return K.leaky_relu(model, 0.2)


def upscaler_skip_con(net, x_nhwc, kernel_size=5, filters=64, upscale_factor=4, unique_names=False):
    """model.py:332-363 ([N,h,w,3] -> [N,h*f,w*f,3]); with unique_names=False it raises like Keras does (sixteen '/conv_pre')"""
    x = x_nhwc.permute(0, 3, 1, 2)
    upscale_times = int(math.log(upscale_factor, 2))
    model = net.prelu(net.conv2d(x, 64, 9, 1))
    upsc_model = model
    for index in range(16):
        model = residual_block(net, model, kernel_size, filters, 1, "res_block/%d" % index if unique_names else "")
    model = net.bn(net.conv2d(model, 64, 3, 1))
    model = upsc_model + model
    for index in range(upscale_times):
        model = upsampling_block(net, model, 3, 224, 2, "upscaling/%d/block" % index if unique_names else "")
    resized_input = M.resize_images_tf1(x, 2 ** upscale_times, "bilinear")
    model = torch.cat([resized_input, model], 1)
    model = torch.tanh(net.conv2d(model, 3, 9, 1))
    return model.permute(0, 2, 3, 1)


# ---- model.py:505-566 ------------------------------------------------------------------------------------------
def same_size_unetish_block(net, model, kernel_size, filters, strides, name, dropout_rate=0.1):
    model = net.conv2d(model, filters, kernel_size, strides, name + "/Conv2D")
    model = net.bn(model)
    model = net.prelu(model, name + "/PReLU")
    return net.dropout(model, dropout_rate, name + "/Dropout")


def upsampling_unetish_block(net, model, kernel_size, filters, strides, name, dropout_rate=0.1):
    model = net.conv2d_transpose(model, filters, kernel_size, strides, name + "/Conv2DTrans")
    model = net.bn(model)
    model = net.prelu(model, name + "/PReLU")
    return net.dropout(model, dropout_rate, name + "/Dropout")


def find_crop_shape(output_down, output_up):
    height_diff, width_diff = output_up.shape[2] - output_down.shape[2], output_up.shape[3] - output_down.shape[3]
    top_crop, left_crop = height_diff // 2, width_diff // 2
    return ((top_crop, height_diff - top_crop), (left_crop, width_diff - left_crop))


def concatenate_layers(output_down, output_up):
    return torch.cat([output_down, Net.crop(output_up, find_crop_shape(output_down, output_up))], 1)


def sum_layers(output_down, output_up):
    return output_down + Net.crop(output_up, find_crop_shape(output_down, output_up))


def _u(net, x, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, join, halve_after_bottom):
    upscale_times = int(math.log(upscale_factor, 2)) + downscale_times
    model = net.prelu(net.conv2d(x, initial_step_filter_count, 9, 1, "initial/Conv2D"), "initial/PReLU")
    outputs, step_filter_count, step = [], initial_step_filter_count, 0
    for step in range(downscale_times):
        for index in range(step_size):
            model = same_size_unetish_block(net, model, kernel_size, step_filter_count, 1, "down/%d/same/%d" % (step, index))      # rate 0.1 (:590)
        outputs.append(model)
        model = same_size_unetish_block(net, model, kernel_size, step_filter_count, 2, "down/%d/down" % step, dropout_rate)
        step_filter_count *= 2
    for index in range(step_size):
        model = same_size_unetish_block(net, model, kernel_size, step_filter_count, 1, "bottom/%d/same/%d" % (step, index), dropout_rate)
    if halve_after_bottom:
        step_filter_count //= 2
    for step in range(upscale_times):
        model = upsampling_unetish_block(net, model, kernel_size, step_filter_count, 2, "up/%d/up" % step, dropout_rate)
        if step < len(outputs):
            model = join(outputs[len(outputs) - step - 1], model)
            step_filter_count //= 2
        for index in range(step_size):
            model = same_size_unetish_block(net, model, kernel_size, step_filter_count, 1, "up/%d/same/%d" % (step, index), dropout_rate)
    return model


def _final_crop(model, out_h, out_w):
    height_diff, width_diff = model.shape[2] - out_h, model.shape[3] - out_w
    top_crop, left_crop = height_diff // 2, width_diff // 2
    return Net.crop(model, ((top_crop, height_diff - top_crop), (left_crop, width_diff - left_crop)))


def upscaler_unetish(net, x_nhwc, kernel_size=5, upscale_factor=4, step_size=4, downscale_times=5, initial_step_filter_count=32, dropout_rate=0.1):
    """model.py:570-634"""
    x = x_nhwc.permute(0, 3, 1, 2)
    model = _u(net, x, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, concatenate_layers, False)
    model = torch.tanh(net.conv2d(model, 3, 9, 1))
    return _final_crop(model, x.shape[2] * upscale_factor, x.shape[3] * upscale_factor).permute(0, 2, 3, 1)


def upscaler_unetish_add(net, x_nhwc, kernel_size=5, upscale_factor=4, step_size=4, downscale_times=5, initial_step_filter_count=48, dropout_rate=0.1):
    """model.py:642-716"""
    x = x_nhwc.permute(0, 3, 1, 2)
    model = _u(net, x, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, sum_layers, True)
    model = torch.tanh(net.conv2d(model, 3, 9, 1))
    model = _final_crop(model, x.shape[2] * upscale_factor, x.shape[3] * upscale_factor)
    resized_input = torch.atanh(0.99999 * M.resize_images_tf1(x, upscale_factor, "bilinear"))
    model = sum_layers(model, resized_input)
    return torch.tanh(net.conv2d(model, 3, 9, 1)).permute(0, 2, 3, 1)


def upscaler_unetish_complex(net, x_nhwc, kernel_size=5, upscale_factor=4, step_size=4, downscale_times=3, initial_step_filter_count=32,
                             dropout_rate=0.1):
    """model.py:743-827"""
    x = x_nhwc.permute(0, 3, 1, 2)
    model = _u(net, x, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, concatenate_layers, False)
    resized_input = M.resize_images_tf1(x, upscale_factor, "bilinear")
    attention = net.conv2d(resized_input, 3, 9, 1, "final/initial/attention")
    for step in range(3):
        p = "final/%d" % step
        attention = torch.cat([resized_input, attention], 1)
        attention = torch.sigmoid(net.conv2d(attention, 3, 9, 1, p + "/attention"))
        model = net.conv2d(model, 3, 9, 1, p + "/Conv2D")
        att_model = attention * model
        model = torch.cat([att_model, model], 1)
        model = torch.tanh(net.conv2d(model, 3, 9, 1, p + "/Conv2D_after_att"))
        if step < 2:
            model = net.dropout(model, dropout_rate, p + "/Dropout")
    return _final_crop(model, x.shape[2] * upscale_factor, x.shape[3] * upscale_factor).permute(0, 2, 3, 1)


# ---- model.py:372-497 ------------------------------------------------------------------------------------------
def inception_mini_resblock(net, model, filters, name, kernel_size, batch_normalisation=True):
    if batch_normalisation:
        model = net.bn(model, name + "/batch_norm")
    model = net.prelu(model, name + "/prelu")
    return net.conv2d(model, filters, tuple(kernel_size), 1, name + "/%dx%d" % (kernel_size[0], kernel_size[1]))


def inception_resblock_3path(net, model, filters, name, kernel_size=3, bn=True):
    gen, k = model, kernel_size
    a = inception_mini_resblock(net, model, int(filters * 0.5), name + "/a/1", (1, 1), bn)
    b = inception_mini_resblock(net, model, int(filters * 0.5), name + "/b/1", (1, 1), bn)
    b = inception_mini_resblock(net, b, int(filters * 0.5), name + "/b/2", (k, k), bn)
    c = inception_mini_resblock(net, model, int(filters * 0.5), name + "/c/1", (1, 1), bn)
    c = inception_mini_resblock(net, c, int(filters * 0.75), name + "/c/2", (k, k), bn)
    c = inception_mini_resblock(net, c, filters, name + "/c/3", (k, k), bn)
    model = net.conv2d(torch.cat([a, b, c], 1), filters, 1, 1, name + "/final/1x1")
    return gen + model


def inception_resblock_2path(net, model, filters, name, kernel_size=7, bn=True):
    gen, k = model, kernel_size
    a = inception_mini_resblock(net, model, int(filters * 0.5), name + "/a/1", (1, 1), bn)
    b = inception_mini_resblock(net, model, int(filters * 0.3), name + "/b/1", (1, 1), bn)
    b = inception_mini_resblock(net, b, int(filters * 0.4), name + "/b/2", (1, k), bn)
    b = inception_mini_resblock(net, b, int(filters * 0.5), name + "/b/3", (k, 1), bn)
    model = net.conv2d(torch.cat([a, b], 1), filters, 1, 1, name + "/final/1x1")
    return gen + model


def upscaler_incep_resnet(net, x_nhwc, filters=64, upscale_factor=4, a_block_type="3path", a_block_num=5, a_block_kernel=3,
                          b_block_type="2path", b_block_num=10, b_block_kernel=7, c_block_type="2path", c_block_num=5, c_block_kernel=3):
    """model.py:443-497"""
    x = x_nhwc.permute(0, 3, 1, 2)
    model = net.conv2d(x, filters, 9, 1, "initial/conv/9x9")
    upsc_model = model
    for tag, btype, num, kern in (("A", a_block_type, a_block_num, a_block_kernel), ("B", b_block_type, b_block_num, b_block_kernel),
                                  ("c", c_block_type, c_block_num, c_block_kernel)):
        for index in range(num):
            if btype == "3path":
                model = inception_resblock_3path(net, model, filters, "inc_res_block/%s/3p/%d" % (tag, index), kern)
            elif btype == "2path":
                model = inception_resblock_2path(net, model, filters, "inc_res_block/%s/2p/%d" % (tag, index), kern)
    model = net.bn(net.conv2d(model, filters, c_block_kernel, 1, "prefinal/conv2d"), "prefinal/batch_norm")
    model = upsc_model + model
    for index in range(int(math.log(upscale_factor, 2))):
        model = upsampling_block(net, model, c_block_kernel, 256, 2, "upscaling/%d/block" % index)
    return torch.tanh(net.conv2d(model, 3, 9, 1, "final/conv")).permute(0, 2, 3, 1)


def generator_cyclegan(net, x_nhwc, filters=64, n_downsample=2, res_block_num=9, upscale_factor=1, norm="instance", kernel_size=3):
    """the north_star's generator shape (no reference counterpart): Conv 9x9 + norm + PReLU; n_downsample x
    [Conv k s2 doubling the channels + norm + PReLU]; residual blocks (model.py:15-27); (n_downsample + log2 f) x [Conv2DTranspose k s2
    halving them, not below `filters`, + norm + PReLU]; Conv 9x9 -> 3 + tanh.  [N,h,w,3] -> [N,h*f,w*f,3]"""
    net.norm = norm
    m = x_nhwc.permute(0, 3, 1, 2)
    m = net.prelu(net.bn(net.conv2d(m, filters, 9, 1, "stem/conv"), "stem/norm"), "stem/prelu")
    ch = filters
    for i in range(n_downsample):
        ch *= 2
        m = net.prelu(net.bn(net.conv2d(m, ch, kernel_size, 2, "down/%d/conv" % i), "down/%d/norm" % i), "down/%d/prelu" % i)
    for i in range(res_block_num):
        m = residual_block(net, m, kernel_size, ch, 1, "res_block/%d" % i)
    for i in range(n_downsample + int(math.log(upscale_factor, 2))):
        ch = max(ch // 2, filters)
        m = net.prelu(net.bn(net.conv2d_transpose(m, ch, kernel_size, 2, "up/%d/conv_transp" % i), "up/%d/norm" % i), "up/%d/prelu" % i)
    return torch.tanh(net.conv2d(m, 3, 9, 1, "head/conv")).permute(0, 2, 3, 1)


def init_weights(fn, in_shape, seed, **kw):
    """run ``fn`` once on zeros with weight creation enabled: {name: float32 array} in creation (= Keras layer) order"""
    import numpy as np
    w = OrderedDict()
    net = Net(w, False, rng=np.random.RandomState(seed))
    with torch.no_grad():
        fn(net, torch.zeros((1,) + tuple(in_shape), dtype=torch.float64), **kw)
    return OrderedDict((k, v.numpy().astype(np.float32)) for k, v in w.items())
