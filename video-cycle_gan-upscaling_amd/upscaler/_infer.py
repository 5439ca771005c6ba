"""Inference-only generator on the bf16-storage kernels (BASELINE.json config C5: "inference-only generator,
batch=32 256->512 bf16, hipGraph-captured per-frame step").

Serves ``make_upscaler_orig`` models (upscaling/upscaler/model.py:267-295) with kernel_size 3, 64 filters and
upscale_factor 2 -- the topology BASELINE.json's configs name.  What the reference does per call is
``upscaler.predict(batch)`` (upscaling/upscaler/data.py:358-363, train_gan3.py:346): a forward pass with the
BatchNormalization layers in inference mode.  Here that pass is

    initial/conv 9x9 3->64 + PReLU            vcg_conv9x9_from3_bf16_fwd (fp32 NCHW frames in, bf16 NHWC out)
    res blocks: conv 3x3 + BN + PReLU         vcg_conv2d_bf16_fwd, BN folded into the epilogue's scale/shift
                conv 3x3 + BN + Add           same kernel, residual operand = block input
    prefinal conv 3x3 + BN + Add(long skip)   same kernel
    upsampling: ConvT 3x3 s2 64->256 + LReLU  vcg_conv_transpose2d_bf16_fwd
    final/conv 9x9 256->3 + tanh              vcg_conv9x9_to3_bf16_fwd (fp32 NCHW out)

22 launches, recorded once per input shape into a hipGraph and replayed per batch.  Activations are bf16 NHWC,
accumulation and the epilogue arithmetic fp32; the weights are rounded to bf16 once, when the engine is built or
``refresh()`` is called after a weight update.  The folded BatchNormalization parameters
(scale = gamma / sqrt(moving_var + 1e-3), shift = (bias - moving_mean) * scale + beta) are 64-element fp32 vectors
derived by vcg_axpby + vcg_norm_finalize when the engine is built or refreshed."""
import ctypes
import os

import numpy as np
import torch

from . import _engine as E
from . import _lib as L

BN_EPS = 1e-3          # keras.layers.BatchNormalization default


class Bf16Generator:
    def __init__(self, model):
        from .model import UpscalerOrig
        if not isinstance(model, UpscalerOrig):
            raise TypeError("Bf16Generator serves make_upscaler_orig models")
        c1 = model.blocks[0][0] if model.blocks else model.c_pre
        if c1.k != 3 or c1.cin != 64 or c1.cout != 64 or model.upscale_times != 1 or model.c_init.cin != 3:
            raise NotImplementedError("bf16 inference is instantiated for kernel_size=3, filters=64, upscale_factor=2 "
                                      "(the topology of BASELINE.json's configs); use model.predict for other shapes")
        self.instance = model.n_pre.norm == "instance"        # per-image statistics cannot be folded: bf16 norm kernels
        self.model = model
        self.rt = model.rt
        self._graphs = {}
        self._bufs = {}
        self.refresh()

    # ---- parameters ------------------------------------------------------------------------------------------
    def _pack3x3(self, conv, transpose):
        rt = self.rt
        out = torch.empty(9, conv.cout, conv.cin, dtype=torch.bfloat16, device=rt.device)
        L.check(rt.lib.vcg_pack_conv_kernel_bf16(conv.ps[conv.name + "/kernel"].data_ptr(), 9, conv.cout, conv.cin, transpose, 0,
                                                 out.data_ptr(), rt.stream), "vcg_pack_conv_kernel_bf16")
        return out

    def _fold(self, conv, norm):
        """scale = gamma / sqrt(moving_var + eps), shift = (bias - moving_mean) * scale + beta, by the same kernels the
        training path uses: vcg_axpby forms (moving_mean - bias), vcg_norm_finalize turns it into scale / shift"""
        rt, ps = self.rt, conv.ps
        c = conv.cout
        if self.instance:          # non-affine instance norm: only the convolution's bias is applied in its epilogue
            return E.filled_like(rt, ps[conv.name + "/bias"], 1.0), ps[conv.name + "/bias"]
        mean = ps[norm.name + "/moving_mean"].clone()
        E.axpby(rt, ps[conv.name + "/bias"], mean, -1.0, 1.0)
        scale, shift = rt.empty(c), rt.empty(c)
        L.check(rt.lib.vcg_norm_finalize(mean.data_ptr(), ps[norm.name + "/moving_variance"].data_ptr(), ps[norm.name + "/gamma"].data_ptr(),
                                         ps[norm.name + "/beta"].data_ptr(), c, 1, BN_EPS, scale.data_ptr(), shift.data_ptr(), None, None, None,
                                         0.0, 0, rt.stream), "vcg_norm_finalize")
        return scale, shift

    def refresh(self):
        """(re)derive the packed bf16 weights and the folded BatchNormalization vectors from the model's parameters"""
        m, rt = self.model, self.rt
        self.trunk = []
        for (c1, n1, c2, n2) in m.blocks:
            s1, h1 = self._fold(c1, n1)
            s2, h2 = self._fold(c2, n2)
            self.trunk.append((self._pack3x3(c1, 1), s1, h1, c1.ps[n1.prelu_name + "/alpha"], self._pack3x3(c2, 1), s2, h2))
        sp, hp = self._fold(m.c_pre, m.n_pre)
        self.prefinal = (self._pack3x3(m.c_pre, 1), sp, hp)
        up = m.ups[0]
        wt = torch.empty(9, up.cout, up.cin, dtype=torch.bfloat16, device=rt.device)
        L.check(rt.lib.vcg_pack_conv_kernel_bf16(up.ps[up.name + "/kernel"].data_ptr(), 9, up.cout, up.cin, 0, 0, wt.data_ptr(),
                                                 rt.stream), "vcg_pack_conv_kernel_bf16")
        self.up = (wt, up.ps[up.name + "/bias"], float(up.alpha))
        wf = torch.empty(L.FINAL9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
        L.check(rt.lib.vcg_pack_final9x9_bf16(m.c_fin.ps[m.c_fin.name + "/kernel"].data_ptr(), wf.data_ptr(), rt.stream),
                "vcg_pack_final9x9_bf16")
        self.final = (wf, m.c_fin.ps[m.c_fin.name + "/bias"])
        w0 = torch.empty(L.FIRST9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
        L.check(rt.lib.vcg_pack_first9x9_bf16(m.c_init.ps[m.c_init.name + "/kernel"].data_ptr(), w0.data_ptr(), rt.stream),
                "vcg_pack_first9x9_bf16")
        self.first = (w0, m.c_init.ps[m.c_init.name + "/bias"], m.c_init.ps[m.a_init.prelu_name + "/alpha"])
        self._graphs.clear()          # recorded graphs hold the old parameter buffers

    # ---- one forward pass (22 launches) -----------------------------------------------------------------------
    def _buffers(self, n, h, w):
        key = (n, h, w)
        if key not in self._bufs:
            dev = self.rt.device
            bf = lambda c, hh, ww: torch.empty(n, hh, ww, c, dtype=torch.bfloat16, device=dev)
            self._bufs[key] = {"skip": bf(64, h, w),
                               "a": bf(64, h, w), "b": bf(64, h, w), "c": bf(64, h, w),
                               "u": torch.empty(self._tail_chunk(n, h, w), 2 * h, 2 * w, 256, dtype=torch.bfloat16, device=dev),
                               "z": bf(64, h, w) if self.instance else None,
                               "stats": torch.empty(5, n * 64, dtype=torch.float32, device=dev) if self.instance else None,
                               "y": torch.empty(n, 3, 2 * h, 2 * w, dtype=torch.float32, device=dev)}
        return self._bufs[key]

    def _tail_chunk(self, n, h, w):
        return E.tail_chunk(n, h, w)

    def _conv(self, x, w, y, scale, shift, act, alpha, res, n, h, wd):
        rt = self.rt
        if self.instance:
            return self._conv_instance_norm(x, w, y, shift, act, alpha, res, n, h, wd)
        d = L.ConvDesc(n, 64, h, wd, 64, h, wd, 3, 3, 1, 1, 1)
        ep = L.EpilogueBf16(scale.data_ptr(), shift.data_ptr(), act, 0.0, alpha.data_ptr() if alpha is not None else None,
                            res.data_ptr() if res is not None else None)
        L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream),
                "vcg_conv2d_bf16_fwd")

    def _conv_instance_norm(self, x, w, y, bias, act, alpha, res, n, h, wd):
        """conv (+bias) -> per-image statistics -> normalise + activation + Add, all on bf16 NHWC"""
        rt = self.rt
        z = self._buffers(n, h, wd)["z"]
        d = L.ConvDesc(n, 64, h, wd, 64, h, wd, 3, 3, 1, 1, 1)
        ep = L.EpilogueBf16(None, bias.data_ptr(), L.ACT_NONE, 0.0, None, None)
        L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), z.data_ptr(), ctypes.byref(ep), rt.stream),
                "vcg_conv2d_bf16_fwd")
        st = self._buffers(n, h, wd)["stats"]
        mean, var, scale, shift, invstd = (st[i] for i in range(5))
        ws, wsn = rt.workspace(rt.lib.vcg_norm_stats_bf16_workspace_bytes(n, 64, h * wd, L.NORM_INSTANCE))
        L.check(rt.lib.vcg_norm_stats_bf16(z.data_ptr(), n, 64, h * wd, L.NORM_INSTANCE, mean.data_ptr(), var.data_ptr(), ws, wsn,
                                           rt.stream), "vcg_norm_stats_bf16")
        L.check(rt.lib.vcg_norm_finalize(mean.data_ptr(), var.data_ptr(), None, None, 64, n, E.IN_EPS, scale.data_ptr(), shift.data_ptr(),
                                         invstd.data_ptr(), None, None, 0.0, 0, rt.stream), "vcg_norm_finalize")
        L.check(rt.lib.vcg_norm_act_fwd_bf16(z.data_ptr(), n, 64, h * wd, scale.data_ptr(), shift.data_ptr(), 1, act, 0.0,
                                             alpha.data_ptr() if alpha is not None else None,
                                             res.data_ptr() if res is not None else None, y.data_ptr(), rt.stream),
                "vcg_norm_act_fwd_bf16")

    def forward(self, x):
        """x: device fp32 NCHW [n,3,h,w] in [-1,1] -> device fp32 NCHW [n,3,2h,2w] (buffer owned by the engine)"""
        rt, m = self.rt, self.model
        n, _, h, w = x.shape
        B = self._buffers(n, h, w)
        w0, b0, a0 = self.first
        d0 = L.ConvDesc(n, 3, h, w, 64, h, w, 9, 9, 1, 4, 4)
        L.check(rt.lib.vcg_conv9x9_from3_bf16_fwd(ctypes.byref(d0), x.data_ptr(), w0.data_ptr(), b0.data_ptr(), a0.data_ptr(),
                                                  B["skip"].data_ptr(), rt.stream), "vcg_conv9x9_from3_bf16_fwd")
        cur = B["skip"]                     # block input; outputs ping-pong between "a" and "c", "skip" is never overwritten
        for (w1, s1, h1, al, w2, s2, h2) in self.trunk:
            out = B["a"] if cur is not B["a"] else B["c"]
            self._conv(cur, w1, B["b"], s1, h1, L.ACT_PRELU, al, None, n, h, w)
            self._conv(B["b"], w2, out, s2, h2, L.ACT_NONE, None, cur, n, h, w)
            cur = out
        wp, sp, hp = self.prefinal
        out = B["a"] if cur is not B["a"] else B["c"]
        self._conv(cur, wp, out, sp, hp, L.ACT_NONE, None, B["skip"], n, h, w)
        wt, bt, slope = self.up
        wf, bf_ = self.final
        ept = L.EpilogueBf16(None, bt.data_ptr(), L.ACT_LRELU, slope, None, None)
        u, y = B["u"], B["y"]
        ch = u.shape[0]
        for i in range(0, n, ch):
            c = min(ch, n - i)
            dt = L.ConvDesc(c, 64, h, w, 256, 2 * h, 2 * w, 3, 3, 2, 0, 0)
            L.check(rt.lib.vcg_conv_transpose2d_bf16_fwd(ctypes.byref(dt), out[i:i + c].data_ptr(), wt.data_ptr(), u.data_ptr(), ctypes.byref(ept),
                                                         rt.stream), "vcg_conv_transpose2d_bf16_fwd")
            df = L.ConvDesc(c, 256, 2 * h, 2 * w, 3, 2 * h, 2 * w, 9, 9, 1, 4, 4)
            L.check(rt.lib.vcg_conv9x9_to3_bf16_fwd(ctypes.byref(df), u.data_ptr(), wf.data_ptr(), bf_.data_ptr(), 1, y[i:i + c].data_ptr(),
                                                    rt.stream), "vcg_conv9x9_to3_bf16_fwd")
        return B["y"]

    # ---- hipGraph ---------------------------------------------------------------------------------------------
    def capture(self, n, h, w):
        """record the pass for one input shape; ``replay(x)`` then costs one graph launch"""
        key = (n, h, w)
        if key in self._graphs:
            return self._graphs[key]
        xin = torch.zeros(n, 3, h, w, dtype=torch.float32, device=self.rt.device)
        self.forward(xin)                  # warm-up: buffers exist, kernel attributes are set
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y = self.forward(xin)
        self._graphs[key] = (g, xin, y)
        return self._graphs[key]

    def replay(self, x):
        g, xin, y = self.capture(*[x.shape[0], x.shape[2], x.shape[3]])
        xin.copy_(x)
        g.replay()
        return y

    # ---- Keras-style entry point -------------------------------------------------------------------------------
    def predict(self, x, batch_size=32):
        """x: numpy NHWC in [-1,1] (data.py:266-270) -> numpy NHWC float32, through the captured graph"""
        rt = self.rt
        x = np.asarray(x)
        outs = []
        for i in range(0, x.shape[0], batch_size):
            xb = E.to_device_nchw(rt, x[i:i + batch_size])
            y = self.replay(xb)
            outs.append(E.to_nhwc(rt, y).cpu().numpy())
        return np.concatenate(outs, 0)
