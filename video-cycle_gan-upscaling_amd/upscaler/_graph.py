"""Functional block API of the reference's ``upscaler.model`` on the device engine.

The reference builds its generators from three block functions operating on Keras tensors:
  residual_block(model, kernel_size, filters, strides, name="")      upscaling/upscaler/model.py:15-27
  downsampling_block(model, kernel_size, filters, strides)           model.py:63-68   (defined, never called)
  upsampling_block(model, kernel_size, filters, strides, name="")    model.py:70-75
This module mirrors them (same names, arguments and layer names) on a small symbolic-tensor graph whose nodes
are the fused device layers of ``_engine`` (Conv2D[+LeakyReLU/tanh], ConvT2D+LeakyReLU, NormAct = BN/IN +
PReLU/LeakyReLU + Add).  ``build_model(inputs, outputs)`` plays the role of ``keras.models.Model(inputs,
outputs)`` and returns a model with the same predict/forward/backward surface as the hand-wired generator, so
custom generators assembled from the reference's blocks train through the same kernels.
"""
import math

import numpy as np

from . import _engine as E
from . import _lib as L


class KTensor:
    """symbolic tensor: (graph, node id, (h, w, c) shape with None for unknown spatial sizes)"""

    def __init__(self, graph, nid, channels):
        self.graph, self.nid, self.channels = graph, nid, channels


class _Graph:
    def __init__(self, input_shape):
        self.input_shape = tuple(input_shape)
        self.nodes = []          # (kind, layer, input nids, attrs)
        self.names = set()

    def add(self, kind, layer, inputs, **attrs):
        if layer is not None:
            if layer.name in self.names:
                raise ValueError("duplicate layer name %r" % layer.name)
            self.names.add(layer.name)
        attrs.pop("name", None)
        self.nodes.append([kind, layer, list(inputs), attrs])
        return len(self.nodes) - 1

    def consumers(self, nid):
        return [i for i, n in enumerate(self.nodes) if nid in n[2]]


_auto = [0]


def _name(name, default):
    if name:
        return name
    _auto[0] += 1
    return "%s_%d" % (default, _auto[0])


def Input(shape, name=None):
    """keras.layers.Input(shape=(h, w, c)): start of a graph (model.py:273)."""
    g = _Graph(shape)
    nid = g.add("input", None, [])
    return KTensor(g, nid, shape[2])


def conv2d(model, filters, kernel_size, strides=1, padding="same", activation=None, name=None):
    """Conv2D(filters, kernel_size, strides, padding) [+ fused LeakyReLU(alpha)/tanh when ``activation`` is
    ('lrelu', alpha) or 'tanh']."""
    act, alpha = L.ACT_NONE, 0.0
    if activation == "tanh":
        act = L.ACT_TANH
    elif isinstance(activation, tuple) and activation[0] == "lrelu":
        act, alpha = L.ACT_LRELU, float(activation[1])
    elif activation is not None:
        raise ValueError("unsupported activation %r" % (activation,))
    layer = E.Conv2D(_name(name, "conv2d"), model.channels, filters, kernel_size, strides, padding, act, alpha)
    return KTensor(model.graph, model.graph.add("conv", layer, [model.nid]), filters)


def batch_norm(model, name=None, norm="batch"):
    layer = E.NormAct(_name(name, "batch_norm"), model.channels, norm)
    return KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), model.channels)


def prelu(model, name=None):
    """PReLU(alpha_initializer='zeros', shared_axes=[1,2]) (model.py:21,276)"""
    n = _name(name, "prelu")
    layer = E.NormAct(n + "_op", model.channels, None, L.ACT_PRELU, prelu_name=n)
    return KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), model.channels)


def leaky_relu(model, alpha, name=None):
    layer = E.NormAct(_name(name, "leaky_relu"), model.channels, None, L.ACT_LRELU, alpha)
    return KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), model.channels)


def add(tensors, name=None):
    """keras.layers.Add()([a, b]).  When b is a freshly made norm node nobody else reads, the addition is
    fused into that node's kernel (y = norm(x) + a)."""
    a, b = tensors
    g = a.graph
    if b.graph is not g:
        raise ValueError("tensors belong to different graphs")
    for x, y in ((a, b), (b, a)):
        node = g.nodes[y.nid]
        if node[0] == "norm" and len(node[2]) == 1 and not g.consumers(y.nid) and node[1].act == L.ACT_NONE \
                and y.nid == len(g.nodes) - 1:
            node[2].append(x.nid)            # second input = residual
            return KTensor(g, y.nid, y.channels)
    layer = E.NormAct(_name(name, "add"), a.channels, None)
    return KTensor(g, g.add("norm", layer, [b.nid, a.nid]), a.channels)


def multiply_sigmoid(attention, model, name=None):
    """Activation('sigmoid')(attention) followed by Multiply()([attention, model]) (model.py:35-36, 88-90): one fused gate"""
    g = model.graph
    if attention.graph is not g or attention.channels != model.channels:
        raise ValueError("attention and model must belong to one graph and have equal channel counts")
    return KTensor(g, g.add("gate", None, [attention.nid, model.nid], name=_name(name, "attention_multiply")), model.channels)


def concatenate(tensors, name=None):
    """Concatenate(axis=3) of NHWC tensors = channel concatenation"""
    g = tensors[0].graph
    return KTensor(g, g.add("concat", None, [t.nid for t in tensors], name=_name(name, "concat")), sum(t.channels for t in tensors))


def resize_images(model, factor, interpolation="nearest", name=None):
    """Lambda(K.resize_images(x, f, f, 'channels_last', interpolation)) (model.py:80-81).  Only factor 1 -- the identity, which is
    what the up-sampling attention block of an x2 generator asks for (scale // 2 with scale = 2) -- is instantiated."""
    if factor != 1:
        raise NotImplementedError("resize_images is instantiated for factor 1 (upscale_factor=2 generators)")
    if interpolation not in ("nearest", "bilinear"):
        raise ValueError(interpolation)
    return model


def atanh_scaled(model, scale=0.99999, name=None):
    """Lambda(lambda x: tf.math.atanh(0.99999 * x)) (model.py:94); defined on data tensors (no gradient)"""
    g = model.graph
    return KTensor(g, g.add("atanh", None, [model.nid], scale=float(scale), name=_name(name, "atanh")), model.channels)


def conv2d_transpose(model, filters, kernel_size, strides=2, activation=None, name=None):
    """Conv2DTranspose(filters, kernel_size, strides=2, padding='same') [+ fused LeakyReLU when activation = ('lrelu', alpha)]"""
    if strides != 2:
        raise NotImplementedError("Conv2DTranspose is implemented for strides=2")
    act, alpha = L.ACT_NONE, 0.0
    if isinstance(activation, tuple) and activation[0] == "lrelu":
        act, alpha = L.ACT_LRELU, float(activation[1])
    elif activation is not None:
        raise ValueError("unsupported activation %r" % (activation,))
    layer = E.ConvT2D(_name(name, "conv_transp"), model.channels, filters, kernel_size, act, alpha)
    return KTensor(model.graph, model.graph.add("convt", layer, [model.nid]), filters)


# ---- the reference's block functions -------------------------------------------------------------------------
def residual_block(model, kernel_size, filters, strides, name="", norm="batch"):
    """model.py:15-27: conv -> BN -> PReLU -> conv -> BN -> Add(block input)."""
    gen = model
    model = conv2d(model, filters, kernel_size, strides, "same", name=name + "/conv_pre")
    n = name + "/prelu"
    layer = E.NormAct(name + "/batch_norm_pre", filters, norm, L.ACT_PRELU, prelu_name=n)
    model = KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), filters)
    model = conv2d(model, filters, kernel_size, strides, "same", name=name + "/conv_post")
    model = batch_norm(model, name=name + "/batch_norm_post", norm=norm)
    return add([gen, model], name=name + "/final_add")


def residual_block_attention(model, input_, kernel_size, filters, strides, batch_norm=True, name="", norm="batch"):
    """model.py:30-48: sigmoid(conv(input_)) gates the block input; conv -> [BN] -> PReLU -> conv -> [BN]; Add(block input)"""
    gen = model
    attention = conv2d(input_, filters, kernel_size, strides, "same", name=name + "/attention")
    model = multiply_sigmoid(attention, model, name=name + "/attention_multiply")
    model = conv2d(model, filters, kernel_size, strides, "same", name=name + "/conv_pre")
    layer = E.NormAct(name + "/batch_norm_pre", filters, norm if batch_norm else None, L.ACT_PRELU, prelu_name=name + "/prelu")
    model = KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), filters)
    model = conv2d(model, filters, kernel_size, strides, "same", name=name + "/conv_post")
    if batch_norm:
        model = batch_norm_(model, name + "/batch_norm_post", norm)
    return add([gen, model], name=name + "/final_add")


def upsampling_block_attention(model, input_, scale, kernel_size, filters, name=""):
    """model.py:78-98: the network input, resized by scale//2 (nearest and bilinear, concatenated), gates the features through a
    sigmoid convolution; Conv2DTranspose(strides 2) + LeakyReLU(0.2); plus Conv2DTranspose(kernel scale+1, strides scale) of
    atanh(0.99999 * input)"""
    if scale != 2:
        raise NotImplementedError("upsampling_block_attention is instantiated for scale=2 (its to_add_input Conv2DTranspose has "
                                  "strides=scale; strides 2 is what the engine implements)")
    near = resize_images(input_, scale // 2, "nearest", name=name + "/nearest")
    bil = resize_images(input_, scale // 2, "bilinear", name=name + "/resize_bilinear")
    up = concatenate([near, bil], name=name + "/upscaled_concat")
    attention = conv2d(up, model.channels, kernel_size, 1, "same", name=name + "/attention")
    model = multiply_sigmoid(attention, model, name=name + "/attention_multiply")
    model = conv2d_transpose(model, filters, kernel_size, 2, activation=("lrelu", 0.2), name=name + "/conv_transp")
    to_add = atanh_scaled(input_, 0.99999, name=name + "/to_add_input_atanh")
    to_add = conv2d_transpose(to_add, filters, scale + 1, scale, name=name + "/to_add_input_conv_transp")
    return add([model, to_add], name=name + "/add_input")


def batch_norm_(model, name, norm):
    return batch_norm(model, name=name, norm=norm)


def downsampling_block(model, kernel_size, filters, strides, name=None):
    """model.py:63-68: Conv2D(strides) + LeakyReLU(0.2)."""
    return conv2d(model, filters, kernel_size, strides, "same", activation=("lrelu", 0.2), name=_name(name, "downsampling"))


def upsampling_block(model, kernel_size, filters, strides, name=""):
    """model.py:70-75: Conv2DTranspose(strides=2, 'same') + LeakyReLU(0.2)."""
    if strides != 2:
        raise NotImplementedError("Conv2DTranspose is implemented for strides=2 (the only value the reference uses)")
    layer = E.ConvT2D(name + "/conv_transp", model.channels, filters, kernel_size, L.ACT_LRELU, 0.2)
    return KTensor(model.graph, model.graph.add("convt", layer, [model.nid]), filters)


# ---- executor ---------------------------------------------------------------------------------------------------
def build_model(inputs, outputs, name="model", seed=7):
    """keras.models.Model(inputs=..., outputs=...) for graphs made with the functions above."""
    from .model import Model

    g = inputs.graph
    if outputs.graph is not g:
        raise ValueError("inputs and outputs belong to different graphs")

    class GraphModel(Model):
        def __init__(self):
            super().__init__(name, g.input_shape, seed)
            self.graph, self.out_nid = g, outputs.nid
            for kind, layer, _, _ in g.nodes:
                if layer is not None:
                    self._add(layer)
            # nodes computed from the network input alone (concat / atanh of the input): data, no gradient flows to them
            self.const = {0}
            for i, (kind, layer, ins, _) in enumerate(g.nodes):
                if kind in ("concat", "atanh") and all(j in self.const for j in ins):
                    self.const.add(i)
            self._finish()

        def _out_shape(self, s):
            h, w = s[0], s[1]
            shapes = {0: (h, w)}
            for i, (kind, layer, ins, _) in enumerate(g.nodes):
                if kind == "input":
                    continue
                ih, iw = shapes[ins[0]]
                if kind == "conv":
                    oh, ow, _, _ = layer.out_hw(ih, iw)
                    shapes[i] = (oh, ow)
                elif kind == "convt":
                    shapes[i] = (2 * ih, 2 * iw)
                elif kind == "gate":
                    shapes[i] = shapes[ins[1]]
                else:
                    shapes[i] = (ih, iw)
            oh, ow = shapes[self.out_nid]
            return (oh, ow, outputs.channels)

        def forward(self, x, training):
            vals, tape = {0: x}, {}
            for i, (kind, layer, ins, _) in enumerate(g.nodes):
                if kind == "input":
                    continue
                if kind in ("conv", "convt"):
                    vals[i], tape[i] = layer.forward(vals[ins[0]])
                elif kind == "gate":
                    a, m = vals[ins[0]], vals[ins[1]]
                    y = self.rt.empty(*m.shape)
                    L.check(self.rt.lib.vcg_sigmoid_gate_fwd(a.data_ptr(), m.data_ptr(), y.data_ptr(), m.numel(), self.rt.stream), "vcg_sigmoid_gate_fwd")
                    vals[i], tape[i] = y, (a, m)
                elif kind == "concat":
                    import torch
                    vals[i], tape[i] = torch.cat([vals[j] for j in ins], 1), None          # NCHW: channel concatenation (memory op)
                elif kind == "atanh":
                    x_ = vals[ins[0]]
                    y = self.rt.empty(*x_.shape)
                    L.check(self.rt.lib.vcg_atanh_scale(x_.data_ptr(), y.data_ptr(), x_.numel(), g.nodes[i][3]["scale"], self.rt.stream), "vcg_atanh_scale")
                    vals[i], tape[i] = y, None
                else:
                    res = vals[ins[1]] if len(ins) > 1 else None
                    vals[i], tape[i] = layer.forward(vals[ins[0]], training, residual=res)
            return vals[self.out_nid], tape

        def backward(self, tape, dy, which=0):
            rt = self.rt
            grads = {self.out_nid: dy}
            for i in range(len(g.nodes) - 1, 0, -1):
                kind, layer, ins, _ = g.nodes[i]
                d = grads.pop(i, None)
                if d is None:
                    continue
                src = ins[0]
                need_dx = src not in self.const
                if kind == "gate":
                    a, m = tape[i]
                    da, dm = rt.empty(*a.shape), rt.empty(*m.shape)
                    L.check(rt.lib.vcg_sigmoid_gate_bwd(a.data_ptr(), m.data_ptr(), d.data_ptr(), da.data_ptr(), dm.data_ptr(), m.numel(),
                                                        rt.stream), "vcg_sigmoid_gate_bwd")
                    self._acc(grads, ins[0], da)
                    self._acc(grads, ins[1], dm)
                    continue
                if kind in ("concat", "atanh"):
                    if i not in self.const:
                        raise NotImplementedError("gradients through %s of non-input tensors" % kind)
                    continue
                if kind == "conv":
                    prev = grads.get(src)
                    dx = layer.backward(tape[i], d, need_dx, True, which, dx_residual=prev)
                    if need_dx:
                        grads[src] = dx
                elif kind == "convt":
                    dx = layer.backward(tape[i], d, need_dx, True, which)
                    if need_dx:
                        self._acc(grads, src, dx)
                else:
                    dx = layer.backward(tape[i], d, True, which)
                    if need_dx:
                        self._acc(grads, src, dx)
                    if len(ins) > 1 and ins[1] not in self.const:
                        self._acc(grads, ins[1], d)       # the Add passes its gradient through unchanged
            return None

        def _acc(self, grads, nid, dx):
            if nid in grads:
                if grads[nid] is dx:
                    return
                # never write into a tensor another branch may still hold: accumulate into the new one
                E.axpby(self.rt, grads[nid], dx, 1.0, 1.0)
            grads[nid] = dx

    return GraphModel()


def make_upscaler_orig_functional(output_image_shape, kernel_size=5, filters=64, upscale_factor=4, res_block_num=16,
                                  norm="batch", seed=7):
    """make_upscaler_orig (model.py:267-295) written exactly as the reference writes it -- block by block on the
    functional API -- instead of the hand-wired ``UpscalerOrig``.  Same layer names, same weights, same kernels."""
    input_image_shape = (output_image_shape[0] // upscale_factor, output_image_shape[1] // upscale_factor, output_image_shape[2])
    upscale_times = int(math.log(upscale_factor, 2))
    upscaler_input = Input(shape=input_image_shape, name="initial/input")
    model = conv2d(upscaler_input, filters, 9, 1, "same", name="initial/conv")
    model = prelu(model, name="initial/prelu")
    upsc_model = model
    for index in range(res_block_num):
        model = residual_block(model, kernel_size, filters, 1, name="res_block/" + str(index), norm=norm)
    model = conv2d(model, 64, kernel_size, 1, "same", name="prefinal/conv2d")
    model = batch_norm(model, name="prefinal/batch_norm", norm=norm)
    model = add([upsc_model, model], name="prefinal/tanh")
    for index in range(upscale_times):
        model = upsampling_block(model, kernel_size, 256, 2, name="upscaling/" + str(index) + "/block")
    model = conv2d(model, 3, 9, 1, "same", activation="tanh", name="final/conv")
    return build_model(upscaler_input, model, name="upscaler_orig_functional", seed=seed)


def make_upscaler_attention(output_image_shape, kernel_size=5, filters=64, upscale_factor=4, res_block_num=16, norm="batch", seed=7):
    """make_upscaler_attention (model.py:299-328) -- the default generator of train_gan3.py (:55 'resnet-att') -- written block by
    block as the reference writes it.  The up-sampling attention block's second Conv2DTranspose has strides = 2**(index+1): the
    engine implements strides 2, i.e. upscale_factor=2 (one block, the configuration BASELINE.json names); other factors raise."""
    if upscale_factor != 2:
        raise NotImplementedError("make_upscaler_attention is instantiated for upscale_factor=2 (Conv2DTranspose strides 4 is not built)")
    input_image_shape = (output_image_shape[0] // upscale_factor, output_image_shape[1] // upscale_factor, output_image_shape[2])
    upscale_times = int(math.log(upscale_factor, 2))
    upscaler_input = Input(shape=input_image_shape, name="initial/input")
    model = conv2d(upscaler_input, filters, 9, 1, "same", name="initial/conv")
    model = prelu(model, name="initial/prelu")
    upsc_model = model
    for index in range(res_block_num):
        model = residual_block_attention(model, upscaler_input, kernel_size, filters, 1, name="res_block/" + str(index), norm=norm)
    model = conv2d(model, filters, kernel_size, 1, "same", name="after_res/conv")
    model = batch_norm(model, name="after_res/batch_norm", norm=norm)
    model = add([upsc_model, model], name="after_res/add")
    for index in range(upscale_times):
        scale = 2 ** (index + 1)
        model = upsampling_block_attention(model, upscaler_input, scale, kernel_size, 128, name="upscaling/" + str(index) + "/block")
    model = conv2d(model, 3, 9, 1, "same", activation="tanh", name="final/conv")
    return build_model(upscaler_input, model, name="upscaler_attention", seed=seed)
