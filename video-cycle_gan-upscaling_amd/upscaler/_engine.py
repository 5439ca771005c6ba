"""Device engine: parameter storage and layer forward/backward on top of the C ABI (include/vcg.h).

torch is used only for device memory (torch.empty), the current HIP stream and host<->device
copies; every arithmetic step is a libvcg_hip.so kernel.  Activations are fp32 NCHW.

Layer <-> reference call sites (upscaling/upscaler/model.py):
  Conv2D            :19,22,275,283,290,839-871     ConvT2D   :72
  NormAct           :20-25,276,284-285,840-841     Dense     :876-884
"""
import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L

import os

# VCG_FUSED_STATS=0: the normalisations' statistics by their own pass over the tensor instead of the producing convolution's epilogue (A/B aid)
FUSED_STATS = os.environ.get("VCG_FUSED_STATS", "1") != "0"
# VCG_FOLD_PREDICT=0: learning-phase-0 passes of the bf16 trunk as conv -> separate normalisation pass (A/B aid)
FOLD_PREDICT = os.environ.get("VCG_FOLD_PREDICT", "1") != "0"
STATS_EPILOGUE_F32 = os.environ.get("VCG_STATS_EPILOGUE_F32", "1") != "0"      # fp32 path: normalisation statistics from the convolutions' epilogues (A/B aid)

BN_EPS = 1e-3          # keras BatchNormalization defaults
BN_MOMENTUM = 0.99
IN_EPS = 1e-5          # instance norm (canonical CycleGAN value; no reference counterpart)


def same_pads(size, k, s):
    """TF SAME: out = ceil(in/s); total = max((out-1)*s+k-in, 0); before = total//2."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


class Runtime:
    """Per-process handle: library, device, grow-only workspace."""

    _inst = None

    def __init__(self):
        self.lib = L.load()
        self.device = L.require_gpu()
        self._ws = None
        self._ws_retired = []   # outgrown workspaces stay allocated: recorded hipGraphs keep writing through their pointers
        self.prof = None      # optional kernel-timing hook set by bench.py

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = Runtime()
        return cls._inst

    @property
    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def workspace(self, nbytes):
        """scratch for partial sums (wgrad / norm / loss reductions), shared by all layers on the stream.  It only grows,
        and an outgrown buffer is never freed: a captured train step or inference graph has its address baked into its
        kernel nodes and would otherwise write into memory the allocator may have handed to a live tensor.  Doubling
        keeps the retired buffers below the size of the current one in total."""
        nbytes = max(int(nbytes), 4096)
        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws is not None:
                self._ws_retired.append(self._ws)
                nbytes = max(nbytes, 2 * self._ws.numel())
            self._ws = torch.empty(nbytes + 4096, dtype=torch.uint8, device=self.device)
        return self._ws.data_ptr(), self._ws.numel()

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def zeros(self, *shape):
        t = self.empty(*shape)
        L.check(self.lib.vcg_fill(t.data_ptr(), t.numel(), 0.0, self.stream), "vcg_fill")
        return t


def _ptr(t):
    return None if t is None else t.data_ptr()


class Timed:
    """Context manager used by bench.py to bracket launches of one kernel family with HIP events
    on the stream the kernels run on."""

    def __init__(self, rt, tag):
        self.rt, self.tag = rt, tag

    def __enter__(self):
        p = self.rt.prof
        if p is not None and self.tag in p.tags:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            self.e0 = None

    def __exit__(self, *a):
        if self.e0 is not None:
            self.e1.record()
            self.rt.prof.events.setdefault(self.tag, []).append((self.e0, self.e1))


# =================================================================================================
# parameter store: one flat fp32 buffer of trainables (+grads), one of non-trainable state
# =================================================================================================
class ParamStore:
    def __init__(self):
        self.specs = OrderedDict()   # name -> (shape, trainable, init)
        self.params = None
        self.state = None
        self.grads = None
        self.grads2 = None
        self.views = {}
        self.gviews = {}
        self.g2views = {}

    def declare(self, name, shape, trainable=True):
        if name in self.specs:
            raise ValueError("duplicate weight " + name)
        self.specs[name] = (tuple(int(s) for s in shape), trainable)

    def materialize(self, rt):
        nt = sum(int(np.prod(s)) for s, t in self.specs.values() if t)
        ns = sum(int(np.prod(s)) for s, t in self.specs.values() if not t)
        self.params = rt.zeros(max(nt, 1))
        self.grads = rt.zeros(max(nt, 1))
        self.grads2 = rt.zeros(max(nt, 1))
        self.state = rt.zeros(max(ns, 1))
        ot = os_ = 0
        self.offsets = {}
        for name, (shape, trainable) in self.specs.items():
            n = int(np.prod(shape))
            if trainable:
                self.views[name] = self.params[ot:ot + n].view(shape)
                self.gviews[name] = self.grads[ot:ot + n].view(shape)
                self.g2views[name] = self.grads2[ot:ot + n].view(shape)
                self.offsets[name] = ot
                ot += n
            else:
                self.views[name] = self.state[os_:os_ + n].view(shape)
                os_ += n
        self.n_trainable, self.n_state = nt, ns

    def __getitem__(self, name):
        return self.views[name]

    def grad(self, name, which=0):
        return (self.gviews if which == 0 else self.g2views)[name]

    def count_params(self):
        return sum(int(np.prod(s)) for s, _ in self.specs.values())

    def set_weights(self, weights):
        """weights: dict name -> array (Keras layouts).  Unknown / missing names raise."""
        for name, arr in weights.items():
            if name not in self.views:
                raise KeyError("unknown weight " + name)
            a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
            if tuple(a.shape) != self.specs[name][0]:
                raise ValueError("shape mismatch for %s: %s vs %s" % (name, a.shape, self.specs[name][0]))
            self.views[name].copy_(torch.from_numpy(a))
        missing = [n for n in self.specs if n not in weights]
        return missing

    def get_weights(self):
        return OrderedDict((n, self.views[n].detach().cpu().numpy().copy()) for n in self.specs)


# =================================================================================================
# initialisers (Keras defaults)
# =================================================================================================
def glorot_uniform(rng, shape, fan_in, fan_out):
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


# =================================================================================================
# layers
# =================================================================================================
class Layer:
    def __init__(self, name):
        self.name = name
        self.rt = None
        self.ps = None

    def bind(self, rt, ps):
        self.rt, self.ps = rt, ps

    def init_weights(self, rng):
        return {}

    def refresh(self):
        """called after the parameters changed (optimizer step / set_weights)"""


class Conv2D(Layer):
    """keras.layers.Conv2D(+fused bias/LeakyReLU/tanh epilogue).  padding: 'same' | 'valid' | int; k: int or (kh, kw)."""

    def __init__(self, name, cin, cout, k, stride=1, padding="same", act=L.ACT_NONE, alpha=0.0):
        super().__init__(name)
        self.kh, self.kw = (int(k[0]), int(k[1])) if isinstance(k, (tuple, list)) else (int(k), int(k))
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, self.kh, stride, padding     # k: the square layers' size
        self.act, self.alpha = act, alpha
        self.wt = None
        self._wt_valid = False

    def declare(self, ps):
        ps.declare(self.name + "/kernel", (self.kh, self.kw, self.cin, self.cout))
        ps.declare(self.name + "/bias", (self.cout,))

    def init_weights(self, rng):
        t = self.kh * self.kw
        return {self.name + "/kernel": glorot_uniform(rng, (self.kh, self.kw, self.cin, self.cout), t * self.cin, t * self.cout),
                self.name + "/bias": np.zeros((self.cout,), np.float32)}

    def out_hw(self, h, w):
        kh, kw, s = self.kh, self.kw, self.stride
        if self.padding == "same":
            oh, pt, _ = same_pads(h, kh, s)
            ow, pl, _ = same_pads(w, kw, s)
        elif self.padding == "valid":
            oh, ow, pt, pl = (h - kh) // s + 1, (w - kw) // s + 1, 0, 0
        else:
            p = int(self.padding)
            oh, ow, pt, pl = (h + 2 * p - kh) // s + 1, (w + 2 * p - kw) // s + 1, p, p
        return oh, ow, pt, pl

    def desc(self, n, h, w):
        oh, ow, pt, pl = self.out_hw(h, w)
        return L.ConvDesc(n, self.cin, h, w, self.cout, oh, ow, self.kh, self.kw, self.stride, pt, pl)

    def refresh(self):
        self._wt_valid = False

    def _wt(self):
        """per-tap transposed kernel (kh,kw,out,in), rebuilt lazily after each parameter change"""
        rt = self.rt
        if self.wt is None:
            self.wt = rt.empty(self.kh * self.kw, self.cout, self.cin)
        if not self._wt_valid:
            L.check(rt.lib.vcg_kernel_transpose(self.ps[self.name + "/kernel"].data_ptr(), self.wt.data_ptr(),
                                                self.kh * self.kw, self.cin, self.cout, rt.stream), "vcg_kernel_transpose")
            self._wt_valid = True
        return self.wt

    def forward(self, x, residual=None, tag=None):
        rt = self.rt
        n, _, h, w = x.shape
        d = self.desc(n, h, w)
        y = rt.empty(n, self.cout, d.oh, d.ow)
        ep = L.Epilogue(_ptr(self.ps[self.name + "/bias"]), self.act, float(self.alpha), None, _ptr(residual))
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv2d_fwd(ctypes.byref(d), x.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(),
                                          y.data_ptr(), ctypes.byref(ep), rt.stream), "vcg_conv2d_fwd[%s]" % self.name)
        return y, (x, y if self.act != L.ACT_NONE else None, d)

    def forward_stats(self, x, instance, tag=None):
        """forward (bias, no activation) whose epilogue leaves the statistics of the normalisation behind the layer: returns
        (y, ctx, (records, records per group, shift)) for NormAct.forward(stats=...), or stats None where the layer's kernel has no
        statistics epilogue (the normalisation then reads y itself)"""
        rt = self.rt
        n, _, h, w = x.shape
        d = self.desc(n, h, w)
        nrec = rt.lib.vcg_conv2d_stats_records(ctypes.byref(d), L.STATS_INSTANCE if instance else L.STATS_BATCH) if STATS_EPILOGUE_F32 else -1
        if nrec <= 0 or self.act != L.ACT_NONE:
            y, ctx = self.forward(x, tag=tag)
            return y, ctx, None
        y = rt.empty(n, self.cout, d.oh, d.ow)
        total = nrec * (n if instance else 1)
        rec = rt.empty(total * 2 * self.cout)
        bias = self.ps[self.name + "/bias"]
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv2d_fwd_stats(ctypes.byref(d), x.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(), bias.data_ptr(),
                                                y.data_ptr(), rec.data_ptr(), rt.stream), "vcg_conv2d_fwd_stats[%s]" % self.name)
        return y, (x, None, d), (rec, nrec, bias)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, dx_residual=None, tag=None):
        rt = self.rt
        x, y, d = ctx
        n = d.n
        db_done = False
        if self.act != L.ACT_NONE:
            # activation backward; the bias gradient (per-channel sum of dz) comes out of the same pass
            dz = rt.empty(*dy.shape)
            db = self.ps.grad(self.name + "/bias", which).data_ptr() if param_grads else None
            ws, wsn = rt.workspace(rt.lib.vcg_act_bwd_workspace_bytes(n, self.cout, d.oh * d.ow))
            L.check(rt.lib.vcg_act_bwd(y.data_ptr(), dy.data_ptr(), n, self.cout, d.oh * d.ow, self.act, float(self.alpha),
                                       None, dz.data_ptr(), None, db, ws, wsn, rt.stream), "vcg_act_bwd[%s]" % self.name)
            dy = dz
            db_done = True
        if param_grads:
            need = rt.lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
            ws, wsn = rt.workspace(need)
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(),
                                                self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                None if db_done else self.ps.grad(self.name + "/bias", which).data_ptr(),
                                                ws, wsn, rt.stream),
                        "vcg_conv2d_wgrad[%s]" % self.name)
        dx = None
        if need_dx:
            dx = rt.empty(n, self.cin, d.h, d.w)
            dd, dyd = d, dy
            if self.stride > 2:
                # sparse_512's stride-3 layers: the data gradient as a stride-1 correlation over the zero-dilated gradient
                s = self.stride
                dyd = rt.empty(n, self.cout, (d.oh - 1) * s + 1, (d.ow - 1) * s + 1)
                L.check(rt.lib.vcg_dilate2d(dy.data_ptr(), dyd.data_ptr(), n * self.cout, d.oh, d.ow, s, rt.stream), "vcg_dilate2d")
                dd = L.ConvDesc(n, self.cin, d.h, d.w, self.cout, dyd.shape[2], dyd.shape[3], self.kh, self.kw, 1, d.pad_top, d.pad_left)
            with Timed(rt, tag and tag + "_dgrad"):
                L.check(rt.lib.vcg_conv2d_dgrad(ctypes.byref(dd), dyd.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(),
                                                self._wt().data_ptr(), dx.data_ptr(), _ptr(dx_residual), rt.stream),
                        "vcg_conv2d_dgrad[%s]" % self.name)
        return dx


class ConvT2D(Layer):
    """keras.layers.Conv2DTranspose(strides=2, padding='same') + fused LeakyReLU (model.py:72-73)."""

    def __init__(self, name, cin, cout, k, act=L.ACT_NONE, alpha=0.0):
        super().__init__(name)
        self.cin, self.cout, self.k, self.act, self.alpha = cin, cout, k, act, alpha
        self.wt = None
        self._wt_valid = False

    def declare(self, ps):
        ps.declare(self.name + "/kernel", (self.k, self.k, self.cout, self.cin))
        ps.declare(self.name + "/bias", (self.cout,))

    def init_weights(self, rng):
        k = self.k
        return {self.name + "/kernel": glorot_uniform(rng, (k, k, self.cout, self.cin), k * k * self.cout, k * k * self.cin),
                self.name + "/bias": np.zeros((self.cout,), np.float32)}

    def desc(self, n, h, w):
        crop = max(self.k - 2, 0) // 2
        return L.ConvDesc(n, self.cin, h, w, self.cout, 2 * h, 2 * w, self.k, self.k, 2, crop, crop)

    def refresh(self):
        self._wt_valid = False

    def _wt(self):
        rt = self.rt
        if self.wt is None:
            self.wt = rt.empty(self.k * self.k, self.cin, self.cout)
        if not self._wt_valid:
            L.check(rt.lib.vcg_kernel_transpose(self.ps[self.name + "/kernel"].data_ptr(), self.wt.data_ptr(),
                                                self.k * self.k, self.cout, self.cin, rt.stream), "vcg_kernel_transpose")
            self._wt_valid = True
        return self.wt

    def forward(self, x, tag=None):
        rt = self.rt
        n, _, h, w = x.shape
        d = self.desc(n, h, w)
        y = rt.empty(n, self.cout, d.oh, d.ow)
        ep = L.Epilogue(_ptr(self.ps[self.name + "/bias"]), self.act, float(self.alpha), None, None)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv_transpose2d_fwd(ctypes.byref(d), x.data_ptr(), self._wt().data_ptr(), y.data_ptr(),
                                                    ctypes.byref(ep), rt.stream), "vcg_conv_transpose2d_fwd[%s]" % self.name)
        return y, (x, y if self.act != L.ACT_NONE else None, d)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, tag=None):
        rt = self.rt
        x, y, d = ctx
        db_done = False
        if self.act != L.ACT_NONE:
            dz = rt.empty(*dy.shape)
            db = self.ps.grad(self.name + "/bias", which).data_ptr() if param_grads else None
            ws, wsn = rt.workspace(rt.lib.vcg_act_bwd_workspace_bytes(d.n, self.cout, d.oh * d.ow))
            L.check(rt.lib.vcg_act_bwd(y.data_ptr(), dy.data_ptr(), d.n, self.cout, d.oh * d.ow, self.act, float(self.alpha),
                                       None, dz.data_ptr(), None, db, ws, wsn, rt.stream), "vcg_act_bwd[%s]" % self.name)
            dy = dz
            db_done = True
        if param_grads:
            need = rt.lib.vcg_conv_transpose2d_wgrad_workspace_bytes(ctypes.byref(d))
            ws, wsn = rt.workspace(need)
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv_transpose2d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(),
                                                          self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                          None if db_done else self.ps.grad(self.name + "/bias", which).data_ptr(),
                                                          ws, wsn, rt.stream), "vcg_conv_transpose2d_wgrad[%s]" % self.name)
        dx = None
        if need_dx:
            dx = rt.empty(d.n, self.cin, d.h, d.w)
            with Timed(rt, tag and tag + "_dgrad"):
                L.check(rt.lib.vcg_conv_transpose2d_dgrad(ctypes.byref(d), dy.data_ptr(),
                                                          self.ps[self.name + "/kernel"].data_ptr(), dx.data_ptr(), None,
                                                          rt.stream), "vcg_conv_transpose2d_dgrad[%s]" % self.name)
        return dx


class ConvTDilated(ConvT2D):
    """keras.layers.Conv2DTranspose(strides=s, padding='same') for any stride, as the data gradient of the stride-1 convolution its
    Keras kernel (kh, kw, out, in) is the HWIO kernel of, evaluated over the zero-dilated input: the ``to_add_input`` branch of the x4
    attention generator (kernel 5, strides 4, model.py:95).  That branch transposes a function of the network INPUT, so only the
    forward pass and the parameter gradients exist here; asking for the input gradient raises."""

    def __init__(self, name, cin, cout, k, stride):
        super().__init__(name, cin, cout, k, L.ACT_NONE, 0.0)
        if k < stride:
            raise NotImplementedError("Conv2DTranspose with kernel_size < strides")
        self.stride = stride
        self._ones = None

    def _descs(self, n, h, w):
        s, k = self.stride, self.k
        hd, wd = (h - 1) * s + 1, (w - 1) * s + 1
        crop = (k - s) // 2                                   # 'same': the full transposed convolution, cropped crop / k - s - crop
        # the virtual convolution: [cout, h*s, w*s] -> [cin, hd, wd], stride 1, pads (crop, crop)
        return hd, wd, L.ConvDesc(n, self.cout, h * s, w * s, self.cin, hd, wd, k, k, 1, crop, crop)

    def _wt(self):
        rt = self.rt
        if self.wt is None:
            self.wt = rt.empty(self.k * self.k, self.cin, self.cout)
        if not self._wt_valid:
            L.check(rt.lib.vcg_kernel_transpose(self.ps[self.name + "/kernel"].data_ptr(), self.wt.data_ptr(), self.k * self.k, self.cout, self.cin,
                                                rt.stream), "vcg_kernel_transpose")
            self._wt_valid = True
        return self.wt

    def forward(self, x, tag=None):
        rt = self.rt
        n, _, h, w = x.shape
        hd, wd, d = self._descs(n, h, w)
        xd = rt.empty(n, self.cin, hd, wd)
        L.check(rt.lib.vcg_dilate2d(x.data_ptr(), xd.data_ptr(), n * self.cin, h, w, self.stride, rt.stream), "vcg_dilate2d")
        z = rt.empty(n, self.cout, d.h, d.w)
        L.check(rt.lib.vcg_conv2d_dgrad(ctypes.byref(d), xd.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(), self._wt().data_ptr(), z.data_ptr(),
                                        None, rt.stream), "vcg_conv2d_dgrad[%s]" % self.name)
        if self._ones is None:
            self._ones = filled_like(rt, self.ps[self.name + "/bias"], 1.0)
        y = rt.empty(*z.shape)                                 # + bias: y = 1 * z + bias[c]
        L.check(rt.lib.vcg_norm_act_fwd(z.data_ptr(), n, self.cout, d.h * d.w, self._ones.data_ptr(), self.ps[self.name + "/bias"].data_ptr(), 0,
                                        L.ACT_NONE, 0.0, None, None, y.data_ptr(), rt.stream), "vcg_norm_act_fwd[%s]" % self.name)
        return y, (xd, d)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, tag=None):
        rt = self.rt
        xd, d = ctx
        if need_dx:
            raise NotImplementedError("Conv2DTranspose(strides=%d): the input gradient is not built (the reference applies it to the network input)"
                                      % self.stride)
        if param_grads:
            ws, wsn = rt.workspace(max(rt.lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d)),
                                       rt.lib.vcg_channel_sum_workspace_bytes(d.n, self.cout, d.h * d.w)))
            # dL/dW[tap][out][in] = sum dy[out][i + tap - crop] * xd[in][i]: the weight gradient of the virtual convolution with x := dy, dy := xd
            L.check(rt.lib.vcg_conv2d_wgrad(ctypes.byref(d), dy.data_ptr(), xd.data_ptr(), self.ps.grad(self.name + "/kernel", which).data_ptr(), None,
                                            ws, wsn, rt.stream), "vcg_conv2d_wgrad[%s]" % self.name)
            L.check(rt.lib.vcg_channel_sum(dy.data_ptr(), d.n, self.cout, d.h * d.w, self.ps.grad(self.name + "/bias", which).data_ptr(), ws, wsn,
                                           rt.stream), "vcg_channel_sum[%s]" % self.name)
        return None


class NormAct(Layer):
    """[BatchNormalization | instance norm | identity] -> [PReLU | LeakyReLU | none] -> [+ residual].

    norm: 'batch' (Keras BN, gamma/beta/moving stats), 'instance' (non-affine), None.
    act: ACT_NONE / ACT_LRELU / ACT_PRELU; PReLU owns a per-channel alpha named ``prelu_name``."""

    def __init__(self, name, c, norm="batch", act=L.ACT_NONE, alpha=0.0, prelu_name=None):
        super().__init__(name)
        self.c, self.norm, self.act, self.alpha, self.prelu_name = c, norm, act, alpha, prelu_name

    def declare(self, ps):
        if self.norm == "batch":
            ps.declare(self.name + "/gamma", (self.c,))
            ps.declare(self.name + "/beta", (self.c,))
            ps.declare(self.name + "/moving_mean", (self.c,), trainable=False)
            ps.declare(self.name + "/moving_variance", (self.c,), trainable=False)
        if self.act == L.ACT_PRELU:
            ps.declare(self.prelu_name + "/alpha", (self.c,))

    def init_weights(self, rng):
        w = {}
        if self.norm == "batch":
            w[self.name + "/gamma"] = np.ones((self.c,), np.float32)
            w[self.name + "/beta"] = np.zeros((self.c,), np.float32)
            w[self.name + "/moving_mean"] = np.zeros((self.c,), np.float32)
            w[self.name + "/moving_variance"] = np.ones((self.c,), np.float32)
        if self.act == L.ACT_PRELU:
            w[self.prelu_name + "/alpha"] = np.zeros((self.c,), np.float32)
        return w

    def _alpha_ptr(self):
        return self.ps[self.prelu_name + "/alpha"].data_ptr() if self.act == L.ACT_PRELU else None

    def needs_stats(self, training):
        """does forward(x, training) compute statistics of x (so that a producer's epilogue may hand them over)"""
        return self.norm is not None and bool(training or self.norm == "instance")

    def forward(self, x, training, residual=None, update_moving=True, stats=None):
        """stats: (records, records per group, shift) from the producing convolution's epilogue (Conv2D.forward_stats) -- one
        vcg_norm_finalize_partials_shifted launch then replaces the statistics pass over x and vcg_norm_finalize"""
        rt, ps = self.rt, self.ps
        lib = rt.lib
        if x.dim() == 4:
            n, c, h, w = x.shape
            hw = h * w
        else:
            n, c = x.shape
            hw = 1
        y = rt.empty(*x.shape)
        saved = None
        if self.norm is None:
            L.check(lib.vcg_norm_act_fwd(x.data_ptr(), n, c, hw, None, None, 0, self.act, float(self.alpha),
                                         self._alpha_ptr(), _ptr(residual), y.data_ptr(), rt.stream), "vcg_norm_act_fwd")
            return y, (x, None, None, (n, c, hw))
        inst = self.norm == "instance"
        rows = n if inst else 1
        mode = L.NORM_INSTANCE if inst else L.NORM_BATCH
        scale, shift, invstd = rt.empty(rows * c), rt.empty(rows * c), rt.empty(rows * c)
        gamma = None if inst else ps[self.name + "/gamma"].data_ptr()
        beta = None if inst else ps[self.name + "/beta"].data_ptr()
        if (training or inst) and stats is not None:
            buf, nrec, kshift = stats
            mean = rt.empty(rows * c)
            mm = mv = None
            if not inst and update_moving:
                mm, mv = ps[self.name + "/moving_mean"].data_ptr(), ps[self.name + "/moving_variance"].data_ptr()
            L.check(lib.vcg_norm_finalize_partials_shifted(buf.data_ptr(), nrec, rows, c, float(hw if inst else n * hw), kshift.data_ptr(), gamma, beta,
                                                           IN_EPS if inst else BN_EPS, mean.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                           invstd.data_ptr(), mm, mv, BN_MOMENTUM, 0 if inst else n * hw, rt.stream),
                    "vcg_norm_finalize_partials_shifted[%s]" % self.name)
            saved = (mean, invstd)
        elif training or inst:
            mean, var = rt.empty(rows * c), rt.empty(rows * c)
            ws, wsn = rt.workspace(lib.vcg_norm_stats_workspace_bytes(n, c, hw, mode))
            L.check(lib.vcg_norm_stats(x.data_ptr(), n, c, hw, mode, mean.data_ptr(), var.data_ptr(), ws, wsn, rt.stream),
                    "vcg_norm_stats[%s]" % self.name)
            mm = mv = None
            if not inst and update_moving:
                mm, mv = ps[self.name + "/moving_mean"].data_ptr(), ps[self.name + "/moving_variance"].data_ptr()
            # Keras' TF backend: fused path (4-D) reports the Bessel-corrected variance to the moving
            # average, the 2-D path (Dense BN) the biased one
            ub = (n * hw) if (x.dim() == 4 and not inst) else 0
            L.check(lib.vcg_norm_finalize(mean.data_ptr(), var.data_ptr(), gamma, beta, c, rows,
                                          IN_EPS if inst else BN_EPS, scale.data_ptr(), shift.data_ptr(), invstd.data_ptr(),
                                          mm, mv, BN_MOMENTUM, ub, rt.stream), "vcg_norm_finalize")
            saved = (mean, invstd)
        else:
            L.check(lib.vcg_norm_finalize(ps[self.name + "/moving_mean"].data_ptr(),
                                          ps[self.name + "/moving_variance"].data_ptr(), gamma, beta, c, 1, BN_EPS,
                                          scale.data_ptr(), shift.data_ptr(), invstd.data_ptr(), None, None, 0.0, 0,
                                          rt.stream), "vcg_norm_finalize")
        L.check(lib.vcg_norm_act_fwd(x.data_ptr(), n, c, hw, scale.data_ptr(), shift.data_ptr(), 1 if inst else 0, self.act,
                                     float(self.alpha), self._alpha_ptr(), _ptr(residual), y.data_ptr(), rt.stream),
                "vcg_norm_act_fwd[%s]" % self.name)
        return y, (x, saved, mode, (n, c, hw))

    def backward(self, ctx, dy, param_grads=True, which=0):
        """returns dx; the residual branch's gradient is dy itself (caller handles it)."""
        rt, ps = self.rt, self.ps
        lib = rt.lib
        x, saved, mode, (n, c, hw) = ctx
        dx = rt.empty(*x.shape)
        dalpha = ps.grad(self.prelu_name + "/alpha", which).data_ptr() if (self.act == L.ACT_PRELU and param_grads) else None
        if self.norm is None:
            ws, wsn = rt.workspace(lib.vcg_act_bwd_workspace_bytes(n, c, hw))
            L.check(lib.vcg_act_bwd(x.data_ptr(), dy.data_ptr(), n, c, hw, self.act, float(self.alpha), self._alpha_ptr(),
                                    dx.data_ptr(), dalpha, None, ws, wsn, rt.stream), "vcg_act_bwd[%s]" % self.name)
            return dx
        if saved is None:
            raise RuntimeError("backward through inference-mode normalisation is not defined")
        mean, invstd = saved
        inst = self.norm == "instance"
        gamma = None if inst else ps[self.name + "/gamma"].data_ptr()
        beta = None if inst else ps[self.name + "/beta"].data_ptr()
        dgamma = dbeta = None
        if not inst and param_grads:
            dgamma = ps.grad(self.name + "/gamma", which).data_ptr()
            dbeta = ps.grad(self.name + "/beta", which).data_ptr()
        ws, wsn = rt.workspace(lib.vcg_norm_act_bwd_workspace_bytes(n, c, hw, mode))
        L.check(lib.vcg_norm_act_bwd(x.data_ptr(), dy.data_ptr(), n, c, hw, mode, mean.data_ptr(), invstd.data_ptr(), gamma,
                                     beta, self.act, float(self.alpha), self._alpha_ptr(), 1, dx.data_ptr(), dgamma, dbeta,
                                     dalpha, ws, wsn, rt.stream), "vcg_norm_act_bwd[%s]" % self.name)
        return dx


class Dense(Layer):
    def __init__(self, name, cin, cout):
        super().__init__(name)
        self.cin, self.cout = cin, cout

    def declare(self, ps):
        ps.declare(self.name + "/kernel", (self.cin, self.cout))
        ps.declare(self.name + "/bias", (self.cout,))

    def init_weights(self, rng):
        return {self.name + "/kernel": glorot_uniform(rng, (self.cin, self.cout), self.cin, self.cout),
                self.name + "/bias": np.zeros((self.cout,), np.float32)}

    def forward(self, x):
        rt = self.rt
        b = x.shape[0]
        y = rt.empty(b, self.cout)
        L.check(rt.lib.vcg_dense_fwd(x.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(),
                                     self.ps[self.name + "/bias"].data_ptr(), y.data_ptr(), b, self.cin, self.cout, rt.stream),
                "vcg_dense_fwd[%s]" % self.name)
        return y, (x,)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0):
        rt = self.rt
        (x,) = ctx
        b = x.shape[0]
        if param_grads:
            L.check(rt.lib.vcg_dense_wgrad(x.data_ptr(), dy.data_ptr(), self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                           self.ps.grad(self.name + "/bias", which).data_ptr(), b, self.cin, self.cout,
                                           rt.stream), "vcg_dense_wgrad[%s]" % self.name)
        dx = None
        if need_dx:
            dx = rt.empty(b, self.cin)
            L.check(rt.lib.vcg_dense_dgrad(dy.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(), dx.data_ptr(), b, self.cin,
                                           self.cout, rt.stream), "vcg_dense_dgrad[%s]" % self.name)
        return dx


# =================================================================================================
# bf16-storage trunk layers (BASELINE.json configs C3/C4: bf16 activations, fp32 master weights / gradients /
# statistics).  Activations are bf16 NHWC torch tensors [n,h,w,64]; parameter names and layouts are those of Conv2D /
# NormAct, so a model built with them exchanges weights with the fp32 one and with the reference.
# =================================================================================================
def to_bf16_nhwc(rt, x_nchw):
    n, c, h, w = x_nchw.shape
    y = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_f32_nchw_to_bf16_nhwc(x_nchw.data_ptr(), y.data_ptr(), n, c, h, w, rt.stream), "vcg_f32_nchw_to_bf16_nhwc")
    return y


def from_bf16_nhwc(rt, x_nhwc):
    n, h, w, c = x_nhwc.shape
    y = rt.empty(n, c, h, w)
    L.check(rt.lib.vcg_bf16_nhwc_to_f32_nchw(x_nhwc.data_ptr(), y.data_ptr(), n, c, h, w, rt.stream), "vcg_bf16_nhwc_to_f32_nchw")
    return y


class Conv3x3Bf16(Layer):
    """Conv2D(64, 3, 'same') on bf16 NHWC: forward and data gradient on vcg_conv2d_bf16_fwd (the latter with the kernel
    packed tap-flipped and transposed), weight / bias gradient on vcg_conv2d_bf16_wgrad (fp32, into the master gradient)."""

    def __init__(self, name, cin=64, cout=64):
        super().__init__(name)
        if cin != 64 or cout != 64:
            raise NotImplementedError("the bf16 trunk convolution is instantiated for 64 -> 64 channels")
        self.cin, self.cout, self.k = cin, cout, 3
        self._wf = self._wd = None
        self._valid = False
        self._group = None              # PackGroup3x3: all of a model's 3x3 kernels re-packed in one launch

    def declare(self, ps):
        ps.declare(self.name + "/kernel", (3, 3, self.cin, self.cout))
        ps.declare(self.name + "/bias", (self.cout,))

    def init_weights(self, rng):
        return {self.name + "/kernel": glorot_uniform(rng, (3, 3, self.cin, self.cout), 9 * self.cin, 9 * self.cout),
                self.name + "/bias": np.zeros((self.cout,), np.float32)}

    def refresh(self):
        self._valid = False
        if self._group is not None:
            self._group.valid = False

    def _packed(self):
        rt = self.rt
        if self._group is not None:
            self._group.ensure()
            return self._wf, self._wd
        if self._wf is None:
            self._wf = torch.empty(9, 64, 64, dtype=torch.bfloat16, device=rt.device)
            self._wd = torch.empty(9, 64, 64, dtype=torch.bfloat16, device=rt.device)
        if not self._valid:
            w = self.ps[self.name + "/kernel"].data_ptr()
            L.check(rt.lib.vcg_pack_conv_kernel_bf16(w, 9, 64, 64, 1, 0, self._wf.data_ptr(), rt.stream), "pack fwd")      # [tap][co][ci]
            L.check(rt.lib.vcg_pack_conv_kernel_bf16(w, 9, 64, 64, 0, 1, self._wd.data_ptr(), rt.stream), "pack dgrad")    # [tap'][ci][co]
            self._valid = True
        return self._wf, self._wd

    def _run(self, x, w, bias, residual=None, stats=None, stats_mode=L.STATS_NONE):
        rt = self.rt
        n, h, wd, _ = x.shape
        y = torch.empty(n, h, wd, 64, dtype=torch.bfloat16, device=rt.device)
        d = L.ConvDesc(n, 64, h, wd, 64, h, wd, 3, 3, 1, 1, 1)
        ep = L.EpilogueBf16(None, bias, L.ACT_NONE, 0.0, None, _ptr(residual), _ptr(stats), stats_mode if stats is not None else L.STATS_NONE)
        L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream),
                "vcg_conv2d_bf16_fwd[%s]" % self.name)
        return y, d

    def forward(self, x, tag=None):
        wf, _ = self._packed()
        with Timed(self.rt, tag):
            y, d = self._run(x, wf, self.ps[self.name + "/bias"].data_ptr())
        return y, (x, d)

    def forward_folded(self, x, norm, residual=None, tag=None, folded=None):
        """learning phase 0: conv + BatchNormalization (moving statistics) [+ PReLU] [+ Add] in ONE launch -- the normalisation is an
        affine map per channel, folded with the bias into the epilogue's scale / shift (vcg_bn_fold); the pre-normalisation tensor is
        never stored.  norm: the NormActBf16 behind this convolution (batch norm).  folded: (scale, shift) already derived (fold_batch)."""
        rt, ps = self.rt, self.ps
        n, h, wd, _ = x.shape
        if folded is not None:
            scale, shift = folded
        else:
            scale, shift = rt.empty(64), rt.empty(64)
            L.check(rt.lib.vcg_bn_fold(ps[self.name + "/bias"].data_ptr(), ps[norm.name + "/moving_mean"].data_ptr(),
                                       ps[norm.name + "/moving_variance"].data_ptr(), ps[norm.name + "/gamma"].data_ptr(),
                                       ps[norm.name + "/beta"].data_ptr(), 64, BN_EPS, scale.data_ptr(), shift.data_ptr(), rt.stream), "vcg_bn_fold")
        wf, _ = self._packed()
        y = torch.empty(n, h, wd, 64, dtype=torch.bfloat16, device=rt.device)
        d = L.ConvDesc(n, 64, h, wd, 64, h, wd, 3, 3, 1, 1, 1)
        ep = L.EpilogueBf16(scale.data_ptr(), shift.data_ptr(), norm.act, float(norm.alpha), norm._alpha_ptr(), _ptr(residual), None, L.STATS_NONE)
        with Timed(rt, tag and residual is not None and tag + "_res" or tag):
            L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream),
                    "vcg_conv2d_bf16_fwd[%s]" % self.name)
        return y

    def forward_stats(self, x, instance, tag=None):
        """forward that also leaves the statistics partials of its output for the normalisation behind it (model.py:20,23,284 in
        training mode): returns (y, ctx, stats) with stats = (records buffer, records per group) or None when the kernel serving
        this shape has no statistics epilogue (the caller then runs the separate statistics pass)"""
        rt = self.rt
        n, h, wd, _ = x.shape
        mode = L.STATS_INSTANCE if instance else L.STATS_BATCH
        d = L.ConvDesc(n, 64, h, wd, 64, h, wd, 3, 3, 1, 1, 1)
        nrec = rt.lib.vcg_conv2d_bf16_stats_records(ctypes.byref(d), mode) if FUSED_STATS else -1
        if nrec <= 0:
            y, ctx = self.forward(x, tag)
            return y, ctx, None
        buf = rt.empty((n if instance else 1) * nrec * 2 * 64)
        wf, _ = self._packed()
        with Timed(rt, tag):
            y, d = self._run(x, wf, self.ps[self.name + "/bias"].data_ptr(), None, buf, mode)
        return y, (x, d), (buf, nrec)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, dx_residual=None, tag=None):
        """dx_residual: a gradient that joins at this layer's input (the block's skip branch), added in the epilogue"""
        rt = self.rt
        x, d = ctx
        if param_grads:
            ws, wsn = rt.workspace(rt.lib.vcg_conv2d_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv2d_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                     self.ps.grad(self.name + "/bias", which).data_ptr(), ws, wsn, rt.stream),
                        "vcg_conv2d_bf16_wgrad[%s]" % self.name)
        if not need_dx:
            return None
        _, wdg = self._packed()
        with Timed(rt, tag and tag + ("_dgrad_res" if dx_residual is not None else "_dgrad")):
            dx, _ = self._run(dy, wdg, None, dx_residual)
        return dx


class PackGroup3x3:
    """the bf16 operand copies (forward + data-gradient pack) of ALL 3x3 64 -> 64 convolutions of a model, re-derived by one launch
    (vcg_pack_conv3x3_c64_bf16_batch) the first time any of them is needed after a parameter change -- the trunk's 2 x 19 launches of 4.7 us
    per optimizer step otherwise"""

    def __init__(self, layers):
        self.layers = list(layers)
        self.valid = False
        self.buf = None
        for l in self.layers:
            l._group = self

    def ensure(self):
        if self.valid:
            return
        rt = self.layers[0].rt
        n = len(self.layers)
        if self.buf is None:
            self.buf = torch.empty(n, 2, 9, 64, 64, dtype=torch.bfloat16, device=rt.device)
            for i, l in enumerate(self.layers):
                l._wf, l._wd = self.buf[i, 0], self.buf[i, 1]
        arr = (ctypes.c_void_p * n)(*[l.ps[l.name + "/kernel"].data_ptr() for l in self.layers])
        L.check(rt.lib.vcg_pack_conv3x3_c64_bf16_batch(arr, n, self.buf.data_ptr(), rt.stream), "vcg_pack_conv3x3_c64_bf16_batch")
        self.valid = True


def fold_batch(rt, pairs):
    """(scale, shift) [len(pairs)][64] of the (Conv3x3Bf16, NormActBf16) pairs of a learning-phase-0 pass: one vcg_bn_fold_batch launch"""
    n = len(pairs)
    out = rt.empty(2, n, 64)
    col = lambda f: (ctypes.c_void_p * n)(*[f(cv, nm) for cv, nm in pairs])
    L.check(rt.lib.vcg_bn_fold_batch(col(lambda cv, nm: cv.ps[cv.name + "/bias"].data_ptr()),
                                     col(lambda cv, nm: cv.ps[nm.name + "/moving_mean"].data_ptr()),
                                     col(lambda cv, nm: cv.ps[nm.name + "/moving_variance"].data_ptr()),
                                     col(lambda cv, nm: cv.ps[nm.name + "/gamma"].data_ptr()),
                                     col(lambda cv, nm: cv.ps[nm.name + "/beta"].data_ptr()), n, 64, BN_EPS, out[0].data_ptr(), out[1].data_ptr(),
                                     rt.stream), "vcg_bn_fold_batch")
    return {id(cv): (out[0, i], out[1, i]) for i, (cv, nm) in enumerate(pairs)}


class Conv2DBf16(Conv2D):
    """keras.layers.Conv2D on bf16 NHWC activations, the discriminators' shapes (3x3 / 4x4, stride 1 / 2, channel counts that
    are multiples of 64 -- what the bf16 weight gradient serves): forward and data gradient on vcg_conv2d_nhwc_bf16_* (weights re-laid out as MFMA operand
    fragments after every optimizer step), weight gradient on vcg_conv2d_nhwc_bf16_wgrad; fp32 master weights, gradients and
    bias.  Same parameter names / layouts as Conv2D: a bf16 model exchanges weights with the fp32 one and with the reference."""

    def __init__(self, name, cin, cout, k, stride=1, padding="same", act=L.ACT_NONE, alpha=0.0):
        if act != L.ACT_NONE:
            raise NotImplementedError("Conv2DBf16 carries no fused activation (the layers it serves are followed by a normalisation)")
        if cin % 64 or cout % 64 or k not in (3, 4) or stride not in (1, 2):
            # vcg_conv2d_nhwc_bf16_fwd/_dgrad take more (5x5, stride 3, multiples of 32); the bf16 weight gradient
            # (bf16_gwgrad.hip: gw_plan) serves 3x3 / 4x4, stride 1 / 2, channel multiples of 64 -- a layer is built only if it can train
            raise NotImplementedError("Conv2DBf16 serves 3x3 / 4x4 kernels, stride 1 / 2, channel counts that are multiples of 64")
        super().__init__(name, cin, cout, k, stride, padding, act, alpha)
        self._wf = self._wd = None
        self._pvalid = False

    def refresh(self):
        super().refresh()
        self._pvalid = False

    def _packed(self):
        rt = self.rt
        t = self.k * self.k
        if self._wf is None:
            self._wf = torch.empty(t * self.cin * self.cout, dtype=torch.bfloat16, device=rt.device)
            self._wd = torch.empty(t * self.cin * self.cout, dtype=torch.bfloat16, device=rt.device)
        if not self._pvalid:
            w = self.ps[self.name + "/kernel"].data_ptr()
            L.check(rt.lib.vcg_pack_conv_frag_bf16_pair(w, t, self.cin, self.cout, self._wf.data_ptr(), self._wd.data_ptr(), rt.stream), "pack fwd + dgrad")
            self._pvalid = True
        return self._wf, self._wd

    def forward(self, x, residual=None, tag=None):
        rt = self.rt
        n, h, w, _ = x.shape
        d = self.desc(n, h, w)
        wf, _ = self._packed()
        y = torch.empty(n, d.oh, d.ow, self.cout, dtype=torch.bfloat16, device=rt.device)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv2d_nhwc_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), self.ps[self.name + "/bias"].data_ptr(),
                                                    L.ACT_NONE, 0.0, y.data_ptr(), rt.stream), "vcg_conv2d_nhwc_bf16_fwd[%s]" % self.name)
        return y, (x, None, d)

    def forward_stats(self, x, instance, tag=None):
        """forward + per-tile statistics partials of the output (see Conv3x3Bf16.forward_stats)"""
        rt = self.rt
        n, h, w, _ = x.shape
        d = self.desc(n, h, w)
        mode = L.STATS_INSTANCE if instance else L.STATS_BATCH
        nrec = rt.lib.vcg_conv2d_nhwc_bf16_stats_records(ctypes.byref(d), mode) if FUSED_STATS else -1
        if nrec <= 0:
            y, ctx = self.forward(x, tag=tag)
            return y, ctx, None
        wf, _ = self._packed()
        y = torch.empty(n, d.oh, d.ow, self.cout, dtype=torch.bfloat16, device=rt.device)
        buf = rt.empty((n if instance else 1) * nrec * 2 * self.cout)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv2d_nhwc_bf16_fwd_stats(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), self.ps[self.name + "/bias"].data_ptr(),
                                                          y.data_ptr(), buf.data_ptr(), rt.stream), "vcg_conv2d_nhwc_bf16_fwd_stats[%s]" % self.name)
        return y, (x, None, d), (buf, nrec)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, dx_residual=None, tag=None, input_lrelu_slope=None):
        """input_lrelu_slope: the layer's input is the OUTPUT of a LeakyReLU with this slope; dx is then multiplied by its derivative, i.e. it
        is the gradient in front of that activation (the producing layer's backward then needs no activation pass)"""
        rt = self.rt
        x, _, d = ctx
        if param_grads:
            ws, wsn = rt.workspace(rt.lib.vcg_conv2d_nhwc_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv2d_nhwc_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(),
                                                          self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                          self.ps.grad(self.name + "/bias", which).data_ptr(), ws, wsn, rt.stream),
                        "vcg_conv2d_nhwc_bf16_wgrad[%s]" % self.name)
        if not need_dx:
            return None
        _, wd = self._packed()
        dx = torch.empty_like(x)
        with Timed(rt, tag and tag + "_dgrad"):
            L.check(rt.lib.vcg_conv2d_nhwc_bf16_dgrad(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(),
                                                      x.data_ptr() if input_lrelu_slope is not None else None, float(input_lrelu_slope or 0.0),
                                                      dx.data_ptr(), rt.stream), "vcg_conv2d_nhwc_bf16_dgrad[%s]" % self.name)
        return dx


class InitialConv9x9Bf16(Conv2D):
    """initial/conv + initial/prelu (model.py:275-276) as the entry into the bf16 layout: Conv2D(64, 9, 'same') + bias + PReLU from the
    fp32 NCHW frames to bf16 NHWC in one launch (vcg_conv9x9_from3_bf16_fwd[_train]: bf16 copies of frames and kernel as MFMA operands,
    fp32 accumulation).  In training the value in front of the PReLU is stored too; backward adds the gradients meeting at the output
    (trunk + long skip, both bf16 NHWC), applies the activation's derivative and the slope gradient in one pass
    (vcg_prelu_bwd_nhwc_bf16_to_bf16) whose bf16 NHWC result feeds the bf16 weight-gradient kernel of the 3-channel layers (vcg_conv3ch_bf16_wgrad).  Declares the
    Conv2D's parameters; the PReLU slope belongs to the NormAct layer behind it (same names as the fp32 model)."""

    def __init__(self, name, cin, cout, k):
        if cin != 3 or cout != 64 or k != 9:
            raise NotImplementedError("the bf16 initial convolution is instantiated for 9x9, 3 -> 64 channels")
        super().__init__(name, cin, cout, k)
        self._wf = None
        self._pvalid = False

    def refresh(self):
        super().refresh()
        self._pvalid = False

    def _packed(self):
        rt = self.rt
        if self._wf is None:
            self._wf = torch.empty(L.FIRST9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
        if not self._pvalid:
            L.check(rt.lib.vcg_pack_first9x9_bf16(self.ps[self.name + "/kernel"].data_ptr(), self._wf.data_ptr(), rt.stream), "vcg_pack_first9x9_bf16")
            self._pvalid = True
        return self._wf

    def forward_prelu(self, x, alpha, training, tag=None):
        """x fp32 NCHW frames, alpha: the PReLU slopes [64] -> (y bf16 NHWC, ctx)"""
        rt = self.rt
        n, _, h, w = x.shape
        d = self.desc(n, h, w)
        wf = self._packed()
        y = torch.empty(n, h, w, self.cout, dtype=torch.bfloat16, device=rt.device)
        bias = self.ps[self.name + "/bias"].data_ptr()
        with Timed(rt, tag):
            if training:
                z = torch.empty_like(y)
                L.check(rt.lib.vcg_conv9x9_from3_bf16_fwd_train(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), bias, alpha.data_ptr(), y.data_ptr(),
                                                                z.data_ptr(), rt.stream), "vcg_conv9x9_from3_bf16_fwd_train")
            else:
                z = None
                L.check(rt.lib.vcg_conv9x9_from3_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), bias, alpha.data_ptr(), y.data_ptr(), rt.stream),
                        "vcg_conv9x9_from3_bf16_fwd")
        return y, (x, z, d)

    def backward_prelu(self, ctx, d1, d2, alpha, dalpha, which=0, tag=None):
        """d1 (+ d2 or None): bf16 NHWC gradients at the PReLU's output; writes the kernel / bias gradients and dalpha [64]"""
        rt, lib = self.rt, self.rt.lib
        x, z, d = ctx
        if z is None:
            raise RuntimeError("backward through an inference-mode forward")
        hw = d.h * d.w
        nrec = lib.vcg_prelu_bwd_nhwc_bf16_records(d.n, hw)
        L.check(min(nrec, 0), "vcg_prelu_bwd_nhwc_bf16_records")
        rec = rt.empty(nrec * self.cout)
        gk, gb = self.ps.grad(self.name + "/kernel", which), self.ps.grad(self.name + "/bias", which)
        need = lib.vcg_conv3ch_bf16_wgrad_workspace_bytes(ctypes.byref(d))
        if need:
            # dz stays bf16 NHWC: the weight gradient multiplies it as an MFMA operand (vcg_conv3ch_bf16_wgrad)
            dz = torch.empty(d.n, d.h, d.w, self.cout, dtype=torch.bfloat16, device=rt.device)
            L.check(lib.vcg_prelu_bwd_nhwc_bf16_to_bf16(d1.data_ptr(), _ptr(d2), z.data_ptr(), alpha.data_ptr(), d.n, self.cout, hw, dz.data_ptr(),
                                                        rec.data_ptr(), rt.stream), "vcg_prelu_bwd_nhwc_bf16_to_bf16")
            L.check(lib.vcg_sum_records(rec.data_ptr(), nrec, self.cout, 1.0, dalpha.data_ptr(), rt.stream), "vcg_sum_records[%s]" % self.name)
            ws, wsn = rt.workspace(need)
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(lib.vcg_conv3ch_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dz.data_ptr(), gk.data_ptr(), gb.data_ptr(), ws, wsn, rt.stream),
                        "vcg_conv3ch_bf16_wgrad[%s]" % self.name)
            return
        # odd widths: the fp32 weight-gradient kernel on the fp32 NCHW form of dz
        dz = rt.empty(d.n, self.cout, d.h, d.w)
        L.check(lib.vcg_prelu_bwd_nhwc_bf16(d1.data_ptr(), _ptr(d2), z.data_ptr(), alpha.data_ptr(), d.n, self.cout, hw, dz.data_ptr(),
                                            rec.data_ptr(), rt.stream), "vcg_prelu_bwd_nhwc_bf16")
        L.check(lib.vcg_sum_records(rec.data_ptr(), nrec, self.cout, 1.0, dalpha.data_ptr(), rt.stream), "vcg_sum_records[%s]" % self.name)
        ws, wsn = rt.workspace(lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
        with Timed(rt, tag and tag + "_wgrad"):
            L.check(lib.vcg_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dz.data_ptr(), gk.data_ptr(), gb.data_ptr(), ws, wsn, rt.stream),
                    "vcg_conv2d_wgrad[%s]" % self.name)


class FirstConvBf16(Conv2D):
    """A critic's first layer -- Conv2D on the 3-channel fp32 NCHW frames -- writing bf16 NHWC directly (vcg_conv3ch_bf16_fwd: bf16 copies of
    frames and kernel as MFMA operands, fp32 accumulation, + bias [+ LeakyReLU]): simple_512 / thin_512 block 1 (3x3 'same', model.py:839;
    its BatchNormalization follows on bf16) and the PatchGAN's 4x4 stride-2 layer + LeakyReLU(0.2).  backward takes the bf16 NHWC gradient
    IN FRONT of the activation (the next layer's data gradient applies the LeakyReLU mask: Conv2DBf16.backward(mask=...)); weight gradient on
    vcg_conv3ch_bf16_wgrad (frames and gradient as bf16 MFMA operands), data gradient on vcg_conv3ch_bf16_dgrad."""

    def __init__(self, name, cin, cout, k, stride=1, padding="same", act=L.ACT_NONE, alpha=0.0):
        if cin != 3 or cout % 64 or cout > 512 or (k, stride) not in ((3, 1), (4, 2)) or act not in (L.ACT_NONE, L.ACT_LRELU):
            raise NotImplementedError("FirstConvBf16 serves Conv2D(64m, 3, strides 1) / Conv2D(64m, 4, strides 2) on 3 input channels")
        super().__init__(name, cin, cout, k, stride, padding, act, alpha)
        self._wf = None
        self._pvalid = False

    def refresh(self):
        super().refresh()
        self._pvalid = False

    def _packed(self):
        rt = self.rt
        if self._wf is None:
            self._wf = torch.empty(rt.lib.vcg_conv3ch_bf16_wfrag_bytes(self.kh, self.kw, self.cout), dtype=torch.uint8, device=rt.device)
        if not self._pvalid:
            L.check(rt.lib.vcg_pack_conv3ch_bf16(self.ps[self.name + "/kernel"].data_ptr(), self.kh, self.kw, self.cout, self._wf.data_ptr(), rt.stream),
                    "vcg_pack_conv3ch_bf16")
            self._pvalid = True
        return self._wf

    def forward(self, x, residual=None, tag=None):
        rt = self.rt
        n, _, h, w = x.shape
        d = self.desc(n, h, w)
        wf = self._packed()
        y = torch.empty(n, d.oh, d.ow, self.cout, dtype=torch.bfloat16, device=rt.device)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv3ch_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), self.ps[self.name + "/bias"].data_ptr(),
                                                float(self.alpha) if self.act == L.ACT_LRELU else 1.0, y.data_ptr(), rt.stream),
                    "vcg_conv3ch_bf16_fwd[%s]" % self.name)
        return y, (x, y, d)

    def backward(self, ctx, dz, need_dx=True, param_grads=True, which=0, dx_residual=None, tag=None):
        """dz: bf16 NHWC gradient in front of the activation"""
        rt = self.rt
        x, _, d = ctx
        dz32 = None
        if param_grads:
            gk, gb = self.ps.grad(self.name + "/kernel", which), self.ps.grad(self.name + "/bias", which)
            need = rt.lib.vcg_conv3ch_bf16_wgrad_workspace_bytes(ctypes.byref(d))
            if need:
                ws, wsn = rt.workspace(need)
                with Timed(rt, tag and tag + "_wgrad"):
                    L.check(rt.lib.vcg_conv3ch_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dz.data_ptr(), gk.data_ptr(), gb.data_ptr(), ws, wsn, rt.stream),
                            "vcg_conv3ch_bf16_wgrad[%s]" % self.name)
            else:       # odd widths: the fp32 weight-gradient kernel on an fp32 NCHW copy of dz
                dz32 = from_bf16_nhwc(rt, dz)
                ws, wsn = rt.workspace(rt.lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
                with Timed(rt, tag and tag + "_wgrad"):
                    L.check(rt.lib.vcg_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dz32.data_ptr(), gk.data_ptr(), gb.data_ptr(), ws, wsn, rt.stream),
                            "vcg_conv2d_wgrad[%s]" % self.name)
        if not need_dx:
            return None
        dx = rt.empty(d.n, self.cin, d.h, d.w)
        need = rt.lib.vcg_conv3ch_bf16_dgrad_workspace_bytes(ctypes.byref(d))
        ws, wsn = rt.workspace(need)
        with Timed(rt, tag and tag + "_dgrad"):
            rc = rt.lib.vcg_conv3ch_bf16_dgrad(ctypes.byref(d), dz.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(), dx.data_ptr(), ws, wsn, rt.stream)
            if rc == L.E_UNSUPPORTED:       # other pads / channel counts: the fp32 data-gradient kernel on the fp32 copy of dz
                dz32 = from_bf16_nhwc(rt, dz) if dz32 is None else dz32
                rc = rt.lib.vcg_conv2d_dgrad(ctypes.byref(d), dz32.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(), self._wt().data_ptr(),
                                             dx.data_ptr(), None, rt.stream)
            L.check(rc, "vcg_conv3ch_bf16_dgrad[%s]" % self.name)
        return dx


class ConvCout1Bf16(Conv2D):
    """Conv2D(1, k) on bf16 NHWC activations -- the 70x70 PatchGAN's last layer (512 -> 1, 4x4, zero padding 1) in the bf16 configs:
    forward, data gradient and weight gradient on vcg_conv2d_cout1_nhwc_bf16_* straight from / to the bf16 NHWC tensor (fp32
    weights, fp32 accumulation, fp32 [n,1,oh,ow] output).  Same parameter names / layouts as Conv2D."""

    def __init__(self, name, cin, cout, k, stride=1, padding="same", act=L.ACT_NONE, alpha=0.0):
        if cout != 1 or stride != 1 or act != L.ACT_NONE or k not in (3, 4) or cin % 8 or cin > 512:
            raise NotImplementedError("ConvCout1Bf16 serves Conv2D(1, 3|4, stride 1) on up to 512 channels (multiples of 8), no activation")
        super().__init__(name, cin, cout, k, stride, padding, act, alpha)

    def forward(self, x, residual=None, tag=None):
        rt = self.rt
        n, h, w, _ = x.shape
        d = self.desc(n, h, w)
        y = rt.empty(n, 1, d.oh, d.ow)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv2d_cout1_nhwc_bf16_fwd(ctypes.byref(d), x.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(),
                                                          self.ps[self.name + "/bias"].data_ptr(), y.data_ptr(), rt.stream),
                    "vcg_conv2d_cout1_nhwc_bf16_fwd[%s]" % self.name)
        return y, (x, None, d)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, dx_residual=None, tag=None):
        rt = self.rt
        x, _, d = ctx
        if param_grads:
            ws, wsn = rt.workspace(rt.lib.vcg_conv2d_cout1_nhwc_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv2d_cout1_nhwc_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(),
                                                                self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                                self.ps.grad(self.name + "/bias", which).data_ptr(), ws, wsn, rt.stream),
                        "vcg_conv2d_cout1_nhwc_bf16_wgrad[%s]" % self.name)
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        with Timed(rt, tag and tag + "_dgrad"):
            L.check(rt.lib.vcg_conv2d_cout1_nhwc_bf16_dgrad(ctypes.byref(d), dy.data_ptr(), self.ps[self.name + "/kernel"].data_ptr(),
                                                            dx.data_ptr(), rt.stream), "vcg_conv2d_cout1_nhwc_bf16_dgrad[%s]" % self.name)
        return dx


def bf16_to_f32(rt, x):
    """flat precision change, shape kept (Flatten of NHWC is its memory order)"""
    y = rt.empty(*x.shape)
    L.check(rt.lib.vcg_bf16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), rt.stream), "vcg_bf16_to_f32")
    return y


def f32_to_bf16(rt, x):
    y = torch.empty(*x.shape, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), rt.stream), "vcg_f32_to_bf16")
    return y


TAIL_CHUNK_MB = int(os.environ.get("VCG_TAIL_CHUNK_MB", "0"))       # 0 (default): whole-batch launches of the up-sampling block and final/conv


def tail_chunk(n, h, w, cout=256):
    """frames per launch of the up-sampling block and final/conv (h, w: the block's INPUT size) -- an experiment kept behind
    VCG_TAIL_CHUNK_MB, OFF by default.  The tensor between the two layers (model.py:288-291) is 2 cout bytes per output pixel -- 134 MB per
    512x512 frame -- and only final/conv reads it in the forward pass: walked one chunk of at most TAIL_CHUNK_MB at a time, final/conv could
    find its input in the 256 MiB Infinity Cache.  Measured (profiles/r03_tail_chunk_ab.txt): it does not pay -- per-frame launches of the
    two kernels take 59 + 60 us against 47 + 42 us per frame in the whole-batch launches (too few tiles per launch to fill 256 CUs evenly),
    C5 4563 against 5149 frames/s, C3's shard 522 against 533."""
    per = 4 * h * w * cout * 2
    budget = TAIL_CHUNK_MB << 20
    if TAIL_CHUNK_MB <= 0 or per > budget:
        return n
    return max(1, min(n, budget // per))


class ConvT3x3Bf16(ConvT2D):
    """upsampling_block (model.py:70-75) with bf16 activations: Conv2DTranspose(3, strides 2) 64 -> 64m + bias + LeakyReLU forward on
    vcg_conv_transpose2d_bf16_fwd (bf16 NHWC in and out).  Backward takes the gradient dz in front of the activation (the bf16 data
    gradient of final/conv applies the LeakyReLU derivative itself): data gradient = the stride-2 convolution of dz with the kernel
    read as (in = out-channels, out = in-channels) on vcg_conv2d_nhwc_bf16_fwd, weight gradient on
    vcg_conv_transpose2d_nhwc_bf16_wgrad, bias gradient = per-channel sum of dz (vcg_norm_stats_bf16's shifted sums)."""

    def __init__(self, name, cin, cout, k, act=L.ACT_NONE, alpha=0.0):
        if k != 3 or cin != 64 or cout % 64:
            raise NotImplementedError("the bf16 transposed convolution is instantiated for 3x3, 64 -> 64m channels")
        super().__init__(name, cin, cout, k, act, alpha)
        self._wp = self._wg = None
        self._pvalid = False

    def refresh(self):
        super().refresh()
        self._pvalid = False

    def _packed(self):
        rt = self.rt
        if self._wp is None:
            self._wp = torch.empty(9, self.cout, self.cin, dtype=torch.bfloat16, device=rt.device)
            self._wg = torch.empty(9 * self.cout * self.cin, dtype=torch.bfloat16, device=rt.device)
        if not self._pvalid:
            w = self.ps[self.name + "/kernel"]
            L.check(rt.lib.vcg_pack_conv_kernel_bf16(w.data_ptr(), 9, self.cout, self.cin, 0, 0, self._wp.data_ptr(), rt.stream), "pack convT")
            # data gradient: Keras' (kh,kw,out,in) kernel is a Conv2D kernel (kh,kw,in'=out,out'=in) of the convolution dz -> dx
            L.check(rt.lib.vcg_pack_conv_frag_bf16(w.data_ptr(), 9, self.cin, self.cout, 0, self._wg.data_ptr(), rt.stream), "pack convT dgrad")
            self._pvalid = True
        return self._wp, self._wg

    def forward(self, x, tag=None, out=None):
        """out: optional preallocated [n, 2h, 2w, cout] bf16 tensor (a slice of a whole-batch tensor when the caller walks the batch in chunks)"""
        rt = self.rt
        n, h, w, _ = x.shape
        wp, _ = self._packed()
        y = out if out is not None else torch.empty(n, 2 * h, 2 * w, self.cout, dtype=torch.bfloat16, device=rt.device)
        d = self.desc(n, h, w)
        ep = L.EpilogueBf16(None, self.ps[self.name + "/bias"].data_ptr(), self.act, float(self.alpha), None, None)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv_transpose2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wp.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream),
                    "vcg_conv_transpose2d_bf16_fwd[%s]" % self.name)
        return y, (x, y, d)

    def backward(self, ctx, dz, need_dx=True, param_grads=True, which=0, tag=None, dz_channel_sums=None):
        """dz: bf16 NHWC gradient in front of the LeakyReLU.  Returns dx as bf16 NHWC.  dz_channel_sums: (records, count) left by the
        kernel that produced dz (FinalConv9x9Bf16.backward) -- the bias gradient then needs no pass over dz."""
        rt, lib = self.rt, self.rt.lib
        x, _, d = ctx
        _, wg = self._packed()
        if param_grads:
            ws, wsn = rt.workspace(lib.vcg_conv_transpose2d_nhwc_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(lib.vcg_conv_transpose2d_nhwc_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dz.data_ptr(),
                                                                 self.ps.grad(self.name + "/kernel", which).data_ptr(), ws, wsn, rt.stream),
                        "vcg_conv_transpose2d_nhwc_bf16_wgrad[%s]" % self.name)
            # bias gradient: sum of dz over (n, h, w)
            if dz_channel_sums is not None:
                rec, nrec = dz_channel_sums
                L.check(lib.vcg_sum_records(rec.data_ptr(), nrec, self.cout, 1.0, self.ps.grad(self.name + "/bias", which).data_ptr(), rt.stream),
                        "vcg_sum_records[%s]" % self.name)
        if param_grads and dz_channel_sums is None:
            # ... = its per-channel mean x count (the shifted sums of vcg_norm_stats_bf16)
            hw = d.oh * d.ow
            mean, var = rt.empty(self.cout), rt.empty(self.cout)
            ws, wsn = rt.workspace(lib.vcg_norm_stats_bf16_workspace_bytes(d.n, self.cout, hw, L.NORM_BATCH))
            L.check(lib.vcg_norm_stats_bf16(dz.data_ptr(), d.n, self.cout, hw, L.NORM_BATCH, mean.data_ptr(), var.data_ptr(), ws, wsn, rt.stream),
                    "vcg_norm_stats_bf16[%s]" % self.name)
            axpby(rt, mean, self.ps.grad(self.name + "/bias", which), float(d.n * hw), 0.0)
        if not need_dx:
            return None
        # dx[ci][i] = sum dz[co][2i + k - crop] W[k][co][ci]: a stride-2 convolution over dz (pad = the crop)
        dd = L.ConvDesc(d.n, self.cout, d.oh, d.ow, self.cin, d.h, d.w, self.k, self.k, 2, d.pad_top, d.pad_left)
        dx = torch.empty_like(x)
        with Timed(rt, tag and tag + "_dgrad"):
            L.check(lib.vcg_conv2d_nhwc_bf16_fwd(ctypes.byref(dd), dz.data_ptr(), wg.data_ptr(), None, L.ACT_NONE, 0.0, dx.data_ptr(), rt.stream),
                    "vcg_conv2d_nhwc_bf16_fwd[%s dgrad]" % self.name)
        return dx


class FinalConv9x9Bf16(Conv2D):
    """final/conv (model.py:290-291): Conv2D(3, 9) + tanh on a bf16 NHWC input with 256 channels, fp32 NCHW output.  Forward on
    vcg_conv9x9_to3_bf16_fwd; data gradient on vcg_conv9x9_to3_bf16_dgrad, which also applies the derivative of the LeakyReLU that
    produced the input (so the result is the gradient in front of that activation); weight gradient on vcg_conv9x9_to3_bf16_wgrad
    (even widths; odd ones take the fp32 kernel on an fp32 NCHW copy of the input)."""

    def __init__(self, name, cin, cout, k, act=L.ACT_TANH):
        if cin != 256 or cout != 3 or k != 9:
            raise NotImplementedError("the bf16 final convolution is instantiated for 9x9, 256 -> 3 channels")
        super().__init__(name, cin, cout, k, 1, "same", act)
        self._wf = self._wd = None
        self._pvalid = False

    def refresh(self):
        super().refresh()
        self._pvalid = False

    def _packed(self):
        rt = self.rt
        if self._wf is None:
            self._wf = torch.empty(L.FINAL9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
            self._wd = torch.empty(4 * L.FIRST9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
        if not self._pvalid:
            w = self.ps[self.name + "/kernel"].data_ptr()
            L.check(rt.lib.vcg_pack_final9x9_bf16(w, self._wf.data_ptr(), rt.stream), "vcg_pack_final9x9_bf16")
            L.check(rt.lib.vcg_pack_conv9x9_3ch_bf16(w, 256, 1, self._wd.data_ptr(), rt.stream), "vcg_pack_conv9x9_3ch_bf16")
            self._pvalid = True
        return self._wf, self._wd

    def forward(self, x, residual=None, tag=None, out=None):
        rt = self.rt
        n, h, w, _ = x.shape
        wf, _ = self._packed()
        d = self.desc(n, h, w)
        y = out if out is not None else rt.empty(n, 3, h, w)
        with Timed(rt, tag):
            L.check(rt.lib.vcg_conv9x9_to3_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), self.ps[self.name + "/bias"].data_ptr(),
                                                    1 if self.act == L.ACT_TANH else 0, y.data_ptr(), rt.stream), "vcg_conv9x9_to3_bf16_fwd")
        return y, (x, y, d)

    def backward(self, ctx, dy, need_dx=True, param_grads=True, which=0, tag=None, input_lrelu_slope=None, want_channel_sums=False):
        """returns dx as bf16 NHWC; with input_lrelu_slope it is already the gradient in front of the LeakyReLU whose output x is.
        want_channel_sums: returns (dx, (records, count)) -- per-channel sums of dx out of the kernel's epilogue (the producing layer's
        bias gradient, ConvT3x3Bf16.backward)"""
        rt = self.rt
        x, y, d = ctx
        n = d.n
        db_done = False
        if self.act != L.ACT_NONE:
            dz = rt.empty(*dy.shape)
            db = self.ps.grad(self.name + "/bias", which).data_ptr() if param_grads else None
            ws, wsn = rt.workspace(rt.lib.vcg_act_bwd_workspace_bytes(n, 3, d.oh * d.ow))
            L.check(rt.lib.vcg_act_bwd(y.data_ptr(), dy.data_ptr(), n, 3, d.oh * d.ow, self.act, 0.0, None, dz.data_ptr(), None, db, ws, wsn,
                                       rt.stream), "vcg_act_bwd[%s]" % self.name)
            dy, db_done = dz, True
        if param_grads and d.w % 2 == 0:
            # bf16 weight gradient straight from the bf16 NHWC input (transposed LDS reads); dz enters as a bf16 MFMA operand
            ws, wsn = rt.workspace(max(rt.lib.vcg_conv9x9_to3_bf16_wgrad_workspace_bytes(ctypes.byref(d)),
                                       rt.lib.vcg_channel_sum_workspace_bytes(n, 3, d.oh * d.ow)))
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv9x9_to3_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                          ws, wsn, rt.stream), "vcg_conv9x9_to3_bf16_wgrad[%s]" % self.name)
            if not db_done:
                L.check(rt.lib.vcg_channel_sum(dy.data_ptr(), n, 3, d.oh * d.ow, self.ps.grad(self.name + "/bias", which).data_ptr(), ws, wsn,
                                               rt.stream), "vcg_channel_sum[%s]" % self.name)
        elif param_grads:
            # odd widths: the fp32 kernel on an fp32 NCHW copy of the input (exact conversion)
            x32 = from_bf16_nhwc(rt, x)
            ws, wsn = rt.workspace(rt.lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
            with Timed(rt, tag and tag + "_wgrad"):
                L.check(rt.lib.vcg_conv2d_wgrad(ctypes.byref(d), x32.data_ptr(), dy.data_ptr(), self.ps.grad(self.name + "/kernel", which).data_ptr(),
                                                None if db_done else self.ps.grad(self.name + "/bias", which).data_ptr(), ws, wsn, rt.stream),
                        "vcg_conv2d_wgrad[%s]" % self.name)
            del x32
        if not need_dx:
            return None
        _, wd = self._packed()
        dx = torch.empty_like(x)
        if want_channel_sums:
            nrec = rt.lib.vcg_conv9x9_to3_bf16_dgrad_chsum_records(ctypes.byref(d))
            L.check(min(nrec, 0), "vcg_conv9x9_to3_bf16_dgrad_chsum_records")
            rec = rt.empty(nrec * self.cin)
            with Timed(rt, tag and tag + "_dgrad"):
                L.check(rt.lib.vcg_conv9x9_to3_bf16_dgrad_chsum(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(),
                                                                x.data_ptr() if input_lrelu_slope is not None else None,
                                                                float(input_lrelu_slope or 0.0), dx.data_ptr(), rec.data_ptr(), rt.stream),
                        "vcg_conv9x9_to3_bf16_dgrad_chsum")
            return dx, (rec, nrec)
        with Timed(rt, tag and tag + "_dgrad"):
            L.check(rt.lib.vcg_conv9x9_to3_bf16_dgrad(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(),
                                                      x.data_ptr() if input_lrelu_slope is not None else None,
                                                      float(input_lrelu_slope or 0.0), dx.data_ptr(), rt.stream), "vcg_conv9x9_to3_bf16_dgrad")
        return dx


class NormActBf16(Layer):
    """NormAct on bf16 NHWC (same parameters / names): statistics, affine and activation arithmetic in fp32."""

    def __init__(self, name, c, norm="batch", act=L.ACT_NONE, alpha=0.0, prelu_name=None):
        super().__init__(name)
        if norm not in ("batch", "instance"):
            raise ValueError(norm)
        self.c, self.norm, self.act, self.alpha, self.prelu_name = c, norm, act, alpha, prelu_name

    declare = NormAct.declare
    init_weights = NormAct.init_weights
    _alpha_ptr = NormAct._alpha_ptr

    def needs_stats(self, training):
        """does forward(x, training) compute statistics of x (so that a producer's epilogue may hand them over)"""
        return bool(training or self.norm == "instance")

    def forward(self, x, training, residual=None, update_moving=True, stats=None):
        """stats: (records, records per group) from the producing convolution's epilogue (Conv3x3Bf16 / Conv2DBf16.forward_stats) --
        one vcg_norm_finalize_partials launch then replaces the statistics pass over x and vcg_norm_finalize"""
        rt, ps, lib = self.rt, self.ps, self.rt.lib
        n, h, w, c = x.shape
        hw = h * w
        inst = self.norm == "instance"
        rows = n if inst else 1
        mode = L.NORM_INSTANCE if inst else L.NORM_BATCH
        scale, shift, invstd = rt.empty(rows * c), rt.empty(rows * c), rt.empty(rows * c)
        gamma = None if inst else ps[self.name + "/gamma"].data_ptr()
        beta = None if inst else ps[self.name + "/beta"].data_ptr()
        saved = None
        if (training or inst) and stats is not None:
            buf, nrec = stats
            mean = rt.empty(rows * c)
            mm = mv = None
            if not inst and update_moving:
                mm, mv = ps[self.name + "/moving_mean"].data_ptr(), ps[self.name + "/moving_variance"].data_ptr()
            L.check(lib.vcg_norm_finalize_partials(buf.data_ptr(), nrec, rows, c, float(hw if inst else n * hw), gamma, beta,
                                                   IN_EPS if inst else BN_EPS, mean.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                   invstd.data_ptr(), mm, mv, BN_MOMENTUM, 0 if inst else n * hw, rt.stream),
                    "vcg_norm_finalize_partials[%s]" % self.name)
            saved = (mean, invstd)
        elif training or inst:
            mean, var = rt.empty(rows * c), rt.empty(rows * c)
            ws, wsn = rt.workspace(lib.vcg_norm_stats_bf16_workspace_bytes(n, c, hw, mode))
            L.check(lib.vcg_norm_stats_bf16(x.data_ptr(), n, c, hw, mode, mean.data_ptr(), var.data_ptr(), ws, wsn, rt.stream),
                    "vcg_norm_stats_bf16[%s]" % self.name)
            mm = mv = None
            if not inst and update_moving:
                mm, mv = ps[self.name + "/moving_mean"].data_ptr(), ps[self.name + "/moving_variance"].data_ptr()
            L.check(lib.vcg_norm_finalize(mean.data_ptr(), var.data_ptr(), gamma, beta, c, rows, IN_EPS if inst else BN_EPS, scale.data_ptr(),
                                          shift.data_ptr(), invstd.data_ptr(), mm, mv, BN_MOMENTUM, 0 if inst else n * hw, rt.stream),
                    "vcg_norm_finalize")
            saved = (mean, invstd)
        else:
            L.check(lib.vcg_norm_finalize(ps[self.name + "/moving_mean"].data_ptr(), ps[self.name + "/moving_variance"].data_ptr(), gamma, beta, c, 1,
                                          BN_EPS, scale.data_ptr(), shift.data_ptr(), invstd.data_ptr(), None, None, 0.0, 0, rt.stream),
                    "vcg_norm_finalize")
        y = torch.empty_like(x)
        L.check(lib.vcg_norm_act_fwd_bf16(x.data_ptr(), n, c, hw, scale.data_ptr(), shift.data_ptr(), 1 if inst else 0, self.act, float(self.alpha),
                                          self._alpha_ptr(), _ptr(residual), y.data_ptr(), rt.stream), "vcg_norm_act_fwd_bf16[%s]" % self.name)
        return y, (x, saved, mode, (n, c, hw))

    def backward(self, ctx, dy, param_grads=True, which=0):
        rt, ps, lib = self.rt, self.ps, self.rt.lib
        x, saved, mode, (n, c, hw) = ctx
        if saved is None:
            raise RuntimeError("backward through inference-mode normalisation is not defined")
        mean, invstd = saved
        inst = self.norm == "instance"
        gamma = None if inst else ps[self.name + "/gamma"].data_ptr()
        beta = None if inst else ps[self.name + "/beta"].data_ptr()
        dgamma = dbeta = dalpha = None
        if param_grads:
            if not inst:
                dgamma, dbeta = ps.grad(self.name + "/gamma", which).data_ptr(), ps.grad(self.name + "/beta", which).data_ptr()
            if self.act == L.ACT_PRELU:
                dalpha = ps.grad(self.prelu_name + "/alpha", which).data_ptr()
        dx = torch.empty_like(x)
        ws, wsn = rt.workspace(lib.vcg_norm_act_bwd_bf16_workspace_bytes(n, c, hw, mode))
        L.check(lib.vcg_norm_act_bwd_bf16(x.data_ptr(), dy.data_ptr(), n, c, hw, mode, mean.data_ptr(), invstd.data_ptr(), gamma, beta, self.act,
                                          float(self.alpha), self._alpha_ptr(), 1, dx.data_ptr(), dgamma, dbeta, dalpha, ws, wsn, rt.stream),
                "vcg_norm_act_bwd_bf16[%s]" % self.name)
        return dx


# =================================================================================================
# layout helpers at the API edge
# =================================================================================================
def to_device_nchw(rt, x):
    """numpy / torch NHWC float array -> device fp32 NCHW (vcg_nhwc_to_nchw)."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=rt.device, dtype=torch.float32).contiguous()
    else:
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        t = torch.from_numpy(a).to(rt.device)
    if t.dim() != 4:
        raise ValueError("expected a 4-D NHWC batch, got shape %s" % (tuple(t.shape),))
    n, h, w, c = t.shape
    out = rt.empty(n, c, h, w)
    L.check(rt.lib.vcg_nhwc_to_nchw(t.data_ptr(), out.data_ptr(), n, h, w, c, rt.stream), "vcg_nhwc_to_nchw")
    return out


def to_nhwc(rt, t):
    """device NCHW -> device NHWC (vcg_nchw_to_nhwc)."""
    n, c, h, w = t.shape
    out = rt.empty(n, h, w, c)
    L.check(rt.lib.vcg_nchw_to_nhwc(t.data_ptr(), out.data_ptr(), n, h, w, c, rt.stream), "vcg_nchw_to_nhwc")
    return out


def maxpool2x2(rt, x):
    """MaxPooling2D((2,2)) forward: fp32 NCHW -> [n,c,h//2,w//2]"""
    n, c, h, w = x.shape
    y = rt.empty(n, c, h // 2, w // 2)
    L.check(rt.lib.vcg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), n, c, h, w, rt.stream), "vcg_maxpool2x2_fwd")
    return y


def maxpool2x2_bwd(rt, x, dy):
    n, c, h, w = x.shape
    dx = rt.empty(n, c, h, w)
    L.check(rt.lib.vcg_maxpool2x2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c, h, w, rt.stream), "vcg_maxpool2x2_bwd")
    return dx


def mean_scalar(rt, t, out=None):
    """device scalar tensor holding mean(t) (vcg_mean_reduce); ``out``: a 1-element fp32 view to write it to."""
    if out is None:
        out = rt.empty(1)
    ws, wsn = rt.workspace(rt.lib.vcg_mean_reduce_workspace_bytes(t.numel()))
    L.check(rt.lib.vcg_mean_reduce(t.data_ptr(), t.numel(), out.data_ptr(), ws, wsn, rt.stream), "vcg_mean_reduce")
    return out


def head_act_fwd(rt, z, kind):
    """discriminator output activation (model.py:885-892); kind: _lib.HEAD_*"""
    y = rt.empty(*z.shape)
    L.check(rt.lib.vcg_head_act_fwd(z.data_ptr(), y.data_ptr(), z.numel(), kind, rt.stream), "vcg_head_act_fwd")
    return y


def head_act_bwd(rt, z, dy, kind):
    dz = rt.empty(*z.shape)
    L.check(rt.lib.vcg_head_act_bwd(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), z.numel(), kind, rt.stream), "vcg_head_act_bwd")
    return dz


def gan_loss(rt, mean_a, mean_b, mean_scale, kind, loss_out, da, ga, db=None, gb=0.0):
    """loss_out = act((mean_a - mean_b) * mean_scale); da[:] = act' * ga, db[:] = act' * gb -- all on the device"""
    L.check(rt.lib.vcg_gan_loss(mean_a.data_ptr(), _ptr(mean_b), float(mean_scale), kind, _ptr(loss_out), _ptr(da),
                                da.numel() if da is not None else 0, float(ga), _ptr(db), db.numel() if db is not None else 0,
                                float(gb), rt.stream), "vcg_gan_loss")


def filled_like(rt, t, value):
    out = rt.empty(*t.shape)
    L.check(rt.lib.vcg_fill(out.data_ptr(), out.numel(), float(value), rt.stream), "vcg_fill")
    return out


def axpby(rt, x, y, a, b):
    """y = a*x + b*y"""
    L.check(rt.lib.vcg_axpby(x.data_ptr(), y.data_ptr(), y.numel(), float(a), float(b), rt.stream), "vcg_axpby")
