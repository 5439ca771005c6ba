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
        self.hw = []             # static (h, w) of every node, as Keras infers it (the reference reads it through Model(...).output_shape)
        self.names = set()
        self.uids = {}           # Keras' per-class counters behind the names of unnamed layers (conv2d_1, batch_normalization_2, p_re_lu_1 ...)

    def auto_name(self, cls):
        self.uids[cls] = self.uids.get(cls, 0) + 1
        return "%s_%d" % (cls, self.uids[cls])

    def add(self, kind, layer, inputs, **attrs):
        lname = layer.name if layer is not None else attrs.get("name")
        if lname is not None:
            if lname in self.names:
                # keras.engine.network: 'The name "..." is used N times in the model. All layer names should be unique.'
                raise ValueError("duplicate layer name %r: all layer names should be unique" % lname)
            self.names.add(lname)
        self.nodes.append([kind, layer, list(inputs), attrs])
        self.hw.append(self._infer_hw(kind, layer, list(inputs), attrs))
        return len(self.nodes) - 1

    def _infer_hw(self, kind, layer, ins, attrs):
        if kind == "input":
            return (self.input_shape[0], self.input_shape[1])
        h, w = self.hw[ins[0]]
        if kind == "conv":
            oh, ow, _, _ = layer.out_hw(h, w)
            return (oh, ow)
        if kind == "convt":
            return (attrs.get("stride", 2) * h, attrs.get("stride", 2) * w)
        if kind == "resize":
            return (h * attrs["factor"], w * attrs["factor"])
        if kind == "crop":
            (t, b), (l, r) = attrs["cropping"]
            return (h - t - b, w - l - r)
        if kind == "gate":
            return self.hw[ins[1]]
        for j in ins[1:]:
            if self.hw[j] != (h, w) and kind in ("norm", "concat"):
                raise ValueError("%s of tensors with different sizes %s / %s" % (kind, (h, w), self.hw[j]))
        return (h, w)

    def consumers(self, nid):
        return [i for i, n in enumerate(self.nodes) if nid in n[2]]


def _name(name, default, model=None):
    """an explicit name, or Keras' automatic one (per-class counter, here scoped to the graph)"""
    if name:
        return name
    g = model.graph if isinstance(model, KTensor) else model
    return g.auto_name(default)


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
    layer = E.Conv2D(_name(name, "conv2d", model), model.channels, filters, kernel_size, strides, padding, act, alpha)
    return KTensor(model.graph, model.graph.add("conv", layer, [model.nid]), filters)


def batch_norm(model, name=None, norm="batch"):
    layer = E.NormAct(_name(name, "batch_normalization", model), model.channels, norm)
    return KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), model.channels)


def batch_norm_prelu(model, name=None, prelu_name=None, norm="batch"):
    """BatchNormalization() followed by PReLU(shared_axes=[1,2]) as one fused node (model.py:508-509,517-518,526-527)"""
    n = _name(name, "batch_normalization", model)
    layer = E.NormAct(n, model.channels, norm, L.ACT_PRELU, prelu_name=_name(prelu_name, "p_re_lu", model))
    return KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), model.channels)


def prelu(model, name=None):
    """PReLU(alpha_initializer='zeros', shared_axes=[1,2]) (model.py:21,276)"""
    n = _name(name, "p_re_lu", model)
    layer = E.NormAct(n + "_op", model.channels, None, L.ACT_PRELU, prelu_name=n)
    return KTensor(model.graph, model.graph.add("norm", layer, [model.nid]), model.channels)


def leaky_relu(model, alpha, name=None):
    layer = E.NormAct(_name(name, "leaky_re_lu", model), model.channels, None, L.ACT_LRELU, alpha)
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
            if g.hw[x.nid] != g.hw[y.nid] or x.channels != y.channels:
                raise ValueError("Add of tensors with different shapes")
            node[2].append(x.nid)            # second input = residual
            return KTensor(g, y.nid, y.channels)
    if a.channels != b.channels:
        raise ValueError("Add of tensors with %d and %d channels" % (a.channels, b.channels))
    layer = E.NormAct(_name(name, "add", g), a.channels, None)
    return KTensor(g, g.add("norm", layer, [b.nid, a.nid]), a.channels)


def multiply_sigmoid(attention, model, name=None):
    """Activation('sigmoid')(attention) followed by Multiply()([attention, model]) (model.py:35-36, 88-90): one fused gate"""
    g = model.graph
    if attention.graph is not g or attention.channels != model.channels:
        raise ValueError("attention and model must belong to one graph and have equal channel counts")
    return KTensor(g, g.add("gate", None, [attention.nid, model.nid], name=_name(name, "multiply", g)), model.channels)


def concatenate(tensors, name=None):
    """Concatenate(axis=3) of NHWC tensors = channel concatenation"""
    g = tensors[0].graph
    return KTensor(g, g.add("concat", None, [t.nid for t in tensors], name=_name(name, "concatenate", g)), sum(t.channels for t in tensors))


def resize_images(model, factor, interpolation="nearest", name=None):
    """Lambda(K.resize_images(x, f, f, 'channels_last', interpolation)) (model.py:80-81,352,705,786): tf.image.resize_* with
    align_corners=False.  Forward only -- the reference resizes nothing but the network input, which is data; factor 1 is the identity."""
    if interpolation not in ("nearest", "bilinear"):
        raise ValueError(interpolation)
    if factor == 1:
        return model
    if factor < 1 or int(factor) != factor:
        raise ValueError("resize factor %r" % (factor,))
    g = model.graph
    return KTensor(g, g.add("resize", None, [model.nid], factor=int(factor), bilinear=interpolation == "bilinear", name=_name(name, "lambda", g)),
                   model.channels)


def cropping2d(model, cropping, name=None):
    """Cropping2D(cropping=((top, bottom), (left, right))) (model.py:552,563,626)"""
    (t, b), (l, r) = cropping
    if min(t, b, l, r) < 0:
        raise ValueError("negative cropping %r" % (cropping,))
    g = model.graph
    if t == b == l == r == 0:
        g.names.add(_name(name, "cropping2d", g))         # the layer exists in Keras (its name is taken); it moves no data
        return model
    return KTensor(g, g.add("crop", None, [model.nid], cropping=((t, b), (l, r)), name=_name(name, "cropping2d", g)), model.channels)


def dropout(model, rate, name=None):
    """Dropout(rate) (model.py:510,519,528): identity at inference and for rates outside (0, 1) (keras.layers.Dropout.call); in the
    learning phase x * mask / (1 - rate) with a fresh mask per step (device-side counter, so a recorded hipGraph redraws it)."""
    g = model.graph
    n = _name(name, "dropout", g)
    return KTensor(g, g.add("dropout", None, [model.nid], rate=float(rate), name=n), model.channels)


def activation(model, kind, name=None):
    """Activation('sigmoid' | 'tanh') as a layer of its own (model.py:799): y = act(x)"""
    if kind not in ("sigmoid", "tanh"):
        raise ValueError("unsupported activation %r" % (kind,))
    g = model.graph
    return KTensor(g, g.add("act", None, [model.nid], head=L.HEAD_KINDS[kind], name=_name(name, "activation", g)), model.channels)


def atanh_scaled(model, scale=0.99999, name=None):
    """Lambda(lambda x: tf.math.atanh(0.99999 * x)) (model.py:94); defined on data tensors (no gradient)"""
    g = model.graph
    return KTensor(g, g.add("atanh", None, [model.nid], scale=float(scale), name=_name(name, "lambda", g)), model.channels)


def conv2d_transpose(model, filters, kernel_size, strides=2, activation=None, name=None):
    """Conv2DTranspose(filters, kernel_size, strides, padding='same') [+ fused LeakyReLU when activation = ('lrelu', alpha)].  Strides
    other than 2 (the x4 attention generator's strides-4 transpose of the input, model.py:95) run as a stride-1 data gradient over the
    zero-dilated tensor and are built for tensors derived from the network input only (no input gradient)."""
    if strides != 2:
        if activation is not None:
            raise NotImplementedError("fused activation on a Conv2DTranspose with strides != 2")
        layer = E.ConvTDilated(_name(name, "conv2d_transpose", model), model.channels, filters, kernel_size, int(strides))
        return KTensor(model.graph, model.graph.add("convt", layer, [model.nid], stride=int(strides)), filters)
    act, alpha = L.ACT_NONE, 0.0
    if isinstance(activation, tuple) and activation[0] == "lrelu":
        act, alpha = L.ACT_LRELU, float(activation[1])
    elif activation is not None:
        raise ValueError("unsupported activation %r" % (activation,))
    layer = E.ConvT2D(_name(name, "conv2d_transpose", model), model.channels, filters, kernel_size, act, alpha)
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
    return conv2d(model, filters, kernel_size, strides, "same", activation=("lrelu", 0.2), name=name)


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
            # nodes computed from the network input alone (resized / concatenated / cropped / atanh'd input): data, no gradient flows to them
            self.const = {0}
            for i, (kind, layer, ins, _) in enumerate(g.nodes):
                if kind in ("concat", "atanh", "resize", "crop") and all(j in self.const for j in ins):
                    self.const.add(i)
            self.has_dropout = any(kind == "dropout" and 0.0 < a["rate"] < 1.0 for kind, _, _, a in g.nodes)
            self._drop_step = None           # device-side step counter of the dropout masks
            self._finish()

        def dropout_masks(self, tape):
            """{Dropout layer name: uint8 NCHW keep-mask of the last training forward} (what a parity check feeds the oracle)"""
            return {g.nodes[i][3]["name"]: t for i, t in tape.items() if g.nodes[i][0] == "dropout" and t is not None}

        def norm_contexts(self, tape):
            """{layer name: (NormAct layer, its saved forward context)} of the last training forward -- what a parity check needs to hand the
            oracle the activation masks the device actually used (a pre-activation within fp32 rounding of 0 may fall on either side)"""
            return {layer.name: (layer, tape[i]) for i, (kind, layer, ins, _) in enumerate(g.nodes) if kind == "norm" and tape.get(i) is not None}

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
                    shapes[i] = (g.nodes[i][3].get("stride", 2) * ih, g.nodes[i][3].get("stride", 2) * iw)
                elif kind == "gate":
                    shapes[i] = shapes[ins[1]]
                elif kind == "resize":
                    shapes[i] = (ih * g.nodes[i][3]["factor"], iw * g.nodes[i][3]["factor"])
                elif kind == "crop":
                    (t, b), (l, r) = g.nodes[i][3]["cropping"]
                    shapes[i] = (ih - t - b, iw - l - r)
                else:
                    shapes[i] = (ih, iw)
            oh, ow = shapes[self.out_nid]
            return (oh, ow, outputs.channels)

        def forward(self, x, training):
            rt = self.rt
            vals, tape = {0: x}, {}
            if training and self.has_dropout:
                import torch
                if self._drop_step is None:
                    self._drop_step = torch.zeros(1, dtype=torch.int64, device=rt.device)
                L.check(rt.lib.vcg_counter_inc(self._drop_step.data_ptr(), rt.stream), "vcg_counter_inc")
            # conv node -> the norm node that reads it (one such consumer: its statistics come from the convolution's epilogue)
            stat_consumer, pending = {}, {}
            for j, (kj, lj, insj, _) in enumerate(g.nodes):
                if kj == "norm" and g.nodes[insj[0]][0] == "conv":
                    stat_consumer[insj[0]] = j if insj[0] not in stat_consumer else None
            stat_consumer = {k: v for k, v in stat_consumer.items() if v is not None}
            for i, (kind, layer, ins, attrs) in enumerate(g.nodes):
                if kind == "input":
                    continue
                if kind in ("conv", "convt"):
                    nj = stat_consumer.get(i)
                    if nj is not None and g.nodes[nj][1].needs_stats(training):
                        # the normalisation behind this convolution takes its statistics from the convolution's epilogue (as the hand-wired models do)
                        vals[i], tape[i], pending[nj] = layer.forward_stats(vals[ins[0]], g.nodes[nj][1].norm == "instance")
                    else:
                        vals[i], tape[i] = layer.forward(vals[ins[0]])
                elif kind == "gate":
                    a, m = vals[ins[0]], vals[ins[1]]
                    y = self.rt.empty(*m.shape)
                    L.check(self.rt.lib.vcg_sigmoid_gate_fwd(a.data_ptr(), m.data_ptr(), y.data_ptr(), m.numel(), self.rt.stream), "vcg_sigmoid_gate_fwd")
                    vals[i], tape[i] = y, (a, m)
                elif kind == "concat":
                    parts = [vals[j] for j in ins]
                    n_, _, h_, w_ = parts[0].shape
                    ctot = sum(p_.shape[1] for p_ in parts)
                    y, off = rt.empty(n_, ctot, h_, w_), 0
                    for p_ in parts:                                       # NCHW: channel concatenation = block copies
                        L.check(rt.lib.vcg_copy_channels(p_.data_ptr(), y.data_ptr(), n_, p_.shape[1], 0, ctot, off, p_.shape[1], h_ * w_, rt.stream),
                                "vcg_copy_channels")
                        off += p_.shape[1]
                    vals[i], tape[i] = y, [p_.shape[1] for p_ in parts]
                elif kind == "act":
                    z = vals[ins[0]]
                    vals[i], tape[i] = E.head_act_fwd(rt, z, attrs["head"]), z
                elif kind == "resize":
                    x_ = vals[ins[0]]
                    n_, c_, h_, w_ = x_.shape
                    f = attrs["factor"]
                    y = rt.empty(n_, c_, h_ * f, w_ * f)
                    L.check(rt.lib.vcg_resize2d(x_.data_ptr(), y.data_ptr(), n_ * c_, h_, w_, f, 1 if attrs["bilinear"] else 0, rt.stream), "vcg_resize2d")
                    vals[i], tape[i] = y, None
                elif kind == "crop":
                    x_ = vals[ins[0]]
                    n_, c_, h_, w_ = x_.shape
                    (t_, b_), (l_, r_) = attrs["cropping"]
                    y = rt.empty(n_, c_, h_ - t_ - b_, w_ - l_ - r_)
                    L.check(rt.lib.vcg_crop2d(x_.data_ptr(), y.data_ptr(), n_ * c_, h_, w_, t_, l_, y.shape[2], y.shape[3], rt.stream), "vcg_crop2d")
                    vals[i], tape[i] = y, (h_, w_)
                elif kind == "dropout":
                    x_ = vals[ins[0]]
                    if training and 0.0 < attrs["rate"] < 1.0:
                        import torch
                        import zlib
                        y = rt.empty(*x_.shape)
                        mask = torch.empty(x_.shape, dtype=torch.uint8, device=rt.device)
                        # per layer, per model seed and per data-parallel rank (each rank drops its own shard independently)
                        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
                        seed = (zlib.crc32(attrs["name"].encode()) << 20) ^ ((self._seed * 0x9E3779B1) & 0xFFFFFFFFFFFF) ^ (rank << 52)
                        L.check(rt.lib.vcg_dropout_fwd(x_.data_ptr(), y.data_ptr(), mask.data_ptr(), x_.numel(), attrs["rate"], seed,
                                                       self._drop_step.data_ptr(), rt.stream), "vcg_dropout_fwd")
                        vals[i], tape[i] = y, mask
                    else:
                        vals[i], tape[i] = x_, None
                elif kind == "atanh":
                    x_ = vals[ins[0]]
                    y = self.rt.empty(*x_.shape)
                    L.check(self.rt.lib.vcg_atanh_scale(x_.data_ptr(), y.data_ptr(), x_.numel(), g.nodes[i][3]["scale"], self.rt.stream), "vcg_atanh_scale")
                    vals[i], tape[i] = y, None
                else:
                    res = vals[ins[1]] if len(ins) > 1 else None
                    vals[i], tape[i] = layer.forward(vals[ins[0]], training, residual=res, stats=pending.pop(i, None))
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
                if kind == "concat":
                    if i in self.const:
                        continue
                    n_, ctot, h_, w_ = d.shape
                    off = 0
                    for j, cj in zip(ins, tape[i]):                       # gradient of a concatenation: its channel blocks
                        if j not in self.const:
                            dj = rt.empty(n_, cj, h_, w_)
                            L.check(rt.lib.vcg_copy_channels(d.data_ptr(), dj.data_ptr(), n_, ctot, off, cj, 0, cj, h_ * w_, rt.stream), "vcg_copy_channels")
                            self._acc(grads, j, dj)
                        off += cj
                    continue
                if kind == "crop":
                    if i in self.const:
                        continue
                    n_, c_, oh_, ow_ = d.shape
                    h_, w_ = tape[i]
                    (t_, _), (l_, _) = g.nodes[i][3]["cropping"]
                    dx = rt.empty(n_, c_, h_, w_)
                    L.check(rt.lib.vcg_pad2d(d.data_ptr(), dx.data_ptr(), n_ * c_, oh_, ow_, t_, l_, h_, w_, rt.stream), "vcg_pad2d")
                    self._acc(grads, src, dx)
                    continue
                if kind == "dropout":
                    if tape[i] is None:
                        self._acc(grads, src, d)
                    else:
                        dx = rt.empty(*d.shape)
                        L.check(rt.lib.vcg_dropout_bwd(d.data_ptr(), tape[i].data_ptr(), dx.data_ptr(), d.numel(), g.nodes[i][3]["rate"], rt.stream),
                                "vcg_dropout_bwd")
                        self._acc(grads, src, dx)
                    continue
                if kind == "act":
                    if src not in self.const:
                        self._acc(grads, src, E.head_act_bwd(rt, tape[i], d, g.nodes[i][3]["head"]))
                    continue
                if kind in ("atanh", "resize"):
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
    """make_upscaler_attention (model.py:299-328) -- the default generator of train_gan3.py (:55 'resnet-att', default -d 4) -- written
    block by block as the reference writes it.  Up-sampling block i resizes the input by 2**i (nearest and bilinear) for its attention
    and adds Conv2DTranspose(kernel 2**(i+1)+1, strides 2**(i+1)) of atanh(0.99999 * input) (E.ConvTDilated for strides 4)."""
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


def make_generator_cyclegan(output_image_shape, filters=64, n_downsample=2, res_block_num=9, upscale_factor=1, norm="instance",
                            kernel_size=3, seed=7):
    """BASELINE.json north_star's literal generator shape ("canonical down-sampling CycleGAN generator";
    the reference itself only ships up-scalers): down-sampling convolutions -> residual blocks -> transposed-convolution up-sampling,

        Conv 9x9 (filters) + norm + PReLU                                   stem   (the reference's own 9x9 ends, model.py:275,290)
        n_downsample x [Conv k s2 (2x channels) + norm + PReLU]             down   (downsampling_block's stride-2 convolution, model.py:63-68)
        res_block_num x residual_block (model.py:15-27) at the widest width
        (n_downsample + log2 upscale_factor) x [Conv2DTranspose k s2 (channels / 2, not below `filters`) + norm + PReLU]   (model.py:70-75)
        Conv 9x9 (3) + tanh                                                 head

    written on the reference's functional block API.  norm: 'instance' (north_star) or 'batch'; PReLU slopes start at 0 (= ReLU).
    upscale_factor 1 maps a frame to a frame (CycleGAN); 2 / 4 make it an up-scaler like the reference's."""
    f = int(upscale_factor)
    if f < 1 or f & (f - 1):
        raise ValueError("upscale_factor must be a power of two")
    in_shape = (output_image_shape[0] // f, output_image_shape[1] // f, output_image_shape[2])
    if in_shape[0] % (1 << n_downsample) or in_shape[1] % (1 << n_downsample):
        raise ValueError("input %s is not divisible by 2**n_downsample" % (in_shape,))
    inp = Input(shape=in_shape, name="stem/input")
    model = conv2d(inp, filters, 9, 1, "same", name="stem/conv")
    model = batch_norm_prelu(model, name="stem/norm", prelu_name="stem/prelu", norm=norm)
    ch = filters
    for i in range(n_downsample):
        ch *= 2
        model = conv2d(model, ch, kernel_size, 2, "same", name="down/%d/conv" % i)
        model = batch_norm_prelu(model, name="down/%d/norm" % i, prelu_name="down/%d/prelu" % i, norm=norm)
    for i in range(res_block_num):
        model = residual_block(model, kernel_size, ch, 1, name="res_block/%d" % i, norm=norm)
    for i in range(n_downsample + int(math.log(f, 2))):
        ch = max(ch // 2, filters)
        model = conv2d_transpose(model, ch, kernel_size, 2, name="up/%d/conv_transp" % i)
        model = batch_norm_prelu(model, name="up/%d/norm" % i, prelu_name="up/%d/prelu" % i, norm=norm)
    model = conv2d(model, 3, 9, 1, "same", activation="tanh", name="head/conv")
    return build_model(inp, model, name="generator_cyclegan", seed=seed)
