import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import generators as OG, models as M, keras_ops as K
from upscaler import _engine as E, model as PM, _lib as L
import test_generators_gpu as T
rt = E.Runtime.get()
orig_n, orig_c, orig_t = E.NormAct.backward, E.Conv2D.backward, E.ConvT2D.backward

def nb(self, ctx, dy, param_grads=True, which=0):
    dx = orig_n(self, ctx, dy, param_grads, which)
    x = ctx[0]
    xr = x.detach().cpu().double().requires_grad_(True)
    if self.norm == "instance":
        z = K.instancenorm(xr)
    elif self.norm == "batch":
        z, _, _ = K.batchnorm(xr, self.ps[self.name + "/gamma"].cpu().double(), self.ps[self.name + "/beta"].cpu().double(), None, None, True)
    else:
        z = xr
    if self.act == L.ACT_PRELU:
        z = K.prelu(z, self.ps[self.prelu_name + "/alpha"].cpu().double())
    (z * dy.cpu().double()).sum().backward()
    e = float((dx.cpu().double() - xr.grad).abs().max() / (xr.grad.abs().max() + 1e-30))
    print("   norm %-34s dx err %.2e  |dy| %.2e shape %s" % (self.name, e, float(dy.abs().max()), tuple(x.shape)))
    return dx

def cb(self, ctx, dy, need_dx=True, param_grads=True, which=0, dx_residual=None, tag=None):
    dx = orig_c(self, ctx, dy, need_dx, param_grads, which, dx_residual, tag)
    x = ctx[0]
    xr = x.detach().cpu().double().requires_grad_(True)
    wk = self.ps[self.name + "/kernel"].cpu().double().requires_grad_(True)
    y = K.conv2d(xr, wk, None, self.stride, self.padding)
    if self.act == L.ACT_TANH:
        y = torch.tanh(y + self.ps[self.name + "/bias"].cpu().double().view(1, -1, 1, 1))
    (y * dy.cpu().double()).sum().backward()
    e_w = float((self.ps.grad(self.name + "/kernel", which).cpu().double() - wk.grad).abs().max() / wk.grad.abs().max())
    e_x = float((dx.cpu().double() - (xr.grad + (dx_residual.cpu().double() if dx_residual is not None else 0))).abs().max() / xr.grad.abs().max()) if dx is not None else -1
    print("   conv %-34s dw err %.2e dx err %.2e" % (self.name, e_w, e_x))
    return dx

def tb(self, ctx, dy, need_dx=True, param_grads=True, which=0, tag=None):
    dx = orig_t(self, ctx, dy, need_dx, param_grads, which, tag)
    x = ctx[0]
    xr = x.detach().cpu().double().requires_grad_(True)
    wk = self.ps[self.name + "/kernel"].cpu().double().requires_grad_(True)
    y = K.conv2d_transpose_same(xr, wk, None, 2)
    (y * dy.cpu().double()).sum().backward()
    e_w = float((self.ps.grad(self.name + "/kernel", which).cpu().double() - wk.grad).abs().max() / wk.grad.abs().max())
    e_x = float((dx.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max())
    print("   convT %-33s dw err %.2e dx err %.2e" % (self.name, e_w, e_x))
    return dx
E.NormAct.backward, E.Conv2D.backward, E.ConvT2D.backward = nb, cb, tb
for norm, f, nd, res in (("instance", 1, 2, 3), ("batch", 2, 1, 2)):
    h, w = 24, 40
    okw = dict(filters=64, n_downsample=nd, res_block_num=res, upscale_factor=f, norm=norm)
    G = PM.make_generator_cyclegan((h * f, w * f, 3), seed=3, **okw)
    gw = T._randomise(OG.init_weights(OG.generator_cyclegan, (h, w, 3), 31, **okw), 32)
    G.set_weights_dict(gw)
    rng = np.random.RandomState(33)
    x = (rng.randint(0, 256, (2, h, w, 3)) / 127.5 - 1).astype(np.float32)
    t = (rng.randint(0, 256, (2, h * f, w * f, 3)) / 127.5 - 1).astype(np.float32)
    print("==", norm, f, nd, res)
    yd, tape = G.forward(E.to_device_nchw(rt, x), True)
    val, dy = PM._pixel_loss(rt, yd, E.to_device_nchw(rt, t), "mse", 1.0)
    G.backward(tape, dy, 0)
