"""GPU parity of every C-ABI kernel family against the CPU oracle (oracle/keras_ops.py) on the same
seeded inputs.  Tolerance: 1e-3 max-norm relative (north_star: "within 1e-3 rel fp32"); the fp32 MFMA
path is an exact fp32 FMA chain so observed errors are ~1e-6 (written to gpurun_out/parity_report.txt)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err, report

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _standalone(rt, layer, seed=0, scale_bias=True):
    from upscaler import _engine as E
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    rng = np.random.RandomState(seed)
    w = layer.init_weights(rng)
    for k in w:
        if k.endswith(("/bias", "/beta", "/alpha")):
            w[k] = rng.uniform(-0.5, 0.5, w[k].shape).astype(np.float32)
        if k.endswith("/gamma"):
            w[k] = rng.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
        if k.endswith("/moving_mean"):
            w[k] = rng.uniform(-0.3, 0.3, w[k].shape).astype(np.float32)
        if k.endswith("/moving_variance"):
            w[k] = rng.uniform(0.5, 2.0, w[k].shape).astype(np.float32)
    ps.set_weights(w)
    return ps, {k: torch.tensor(v, dtype=torch.float64) for k, v in w.items()}


CONV_CASES = [
    # cin, cout, k, stride, padding, n, h, w
    (64, 64, 3, 1, "same", 2, 16, 32),
    (256, 256, 3, 1, "same", 2, 6, 10),      # make_generator_cyclegan's residual blocks at h/4
    (128, 256, 3, 2, "same", 2, 12, 20),
    (64, 64, 3, 1, "same", 1, 13, 45),       # ragged tile edges
    (8, 64, 3, 1, "same", 1, 9, 33),
    (64, 128, 3, 2, "same", 2, 16, 32),
    (64, 128, 3, 2, "same", 1, 15, 31),      # odd size: TF SAME pads (1,1)
    (128, 256, 3, 2, "same", 1, 8, 8),
    (512, 512, 3, 2, "same", 2, 4, 4),
    (512, 512, 3, 2, "same", 2, 2, 2),
    (3, 64, 3, 1, "same", 2, 16, 40),
    (3, 64, 9, 1, "same", 1, 16, 40),
    (64, 64, 5, 1, "same", 1, 12, 34),
    (256, 3, 9, 1, "same", 1, 20, 70),       # small-M kernel, 2 x-tiles
    (64, 3, 9, 1, "same", 2, 9, 50),
    (3, 64, 4, 2, 1, 1, 32, 32),             # PatchGAN
    (64, 128, 4, 2, 1, 1, 16, 16),
    (256, 512, 4, 1, 1, 1, 9, 9),
    (512, 1, 4, 1, 1, 1, 10, 10),
    (16, 1, 4, 1, 1, 2, 8, 70),
    (256, 512, 3, 2, "same", 4, 32, 32),     # D block 4 at 128x128 frames, batch 4
    (64, 128, 3, 2, "same", 4, 128, 128),    # D block 2
    (64, 64, 3, 1, "same", 3, 40, 72),
    (128, 256, 4, 2, 1, 2, 32, 32),
    (3, 64, 4, 2, 1, 2, 64, 64),
    (256, 3, 9, 1, "same", 2, 40, 72),
    (256, 3, 9, 1, "same", 1, 20, 64),       # width % 64 == 0: the row-chain kernel (conv_rowchain.hip)
    (256, 3, 9, 1, "same", 2, 70, 128),      # two strips, several row segments
    (256, 1, 9, 1, "same", 1, 9, 64),
    # the inception-resnet generator's shapes (model.py:372-440): 1x1, 1xk / kx1, odd channel counts
    (64, 19, (1, 1), 1, "same", 2, 13, 45),
    (19, 25, (1, 7), 1, "same", 2, 13, 45),
    (25, 32, (7, 1), 1, "same", 2, 13, 45),
    (128, 64, (1, 1), 1, "same", 1, 20, 36),
    (19, 25, (1, 3), 1, "same", 1, 9, 33),
    (25, 32, (3, 1), 1, "same", 1, 9, 33),
    (19, 25, (1, 5), 1, "same", 3, 16, 32),
    (25, 300, (5, 1), 1, "same", 1, 16, 40),   # more than one weight-gradient column block (256 / 5 channels each)
]


@pytest.mark.parametrize("cin,cout,k,stride,padding,n,h,w", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(rt, cin, cout, k, stride, padding, n, h, w):
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    layer = E.Conv2D("c", cin, cout, k, stride, padding)
    ps, wd = _standalone(rt, layer, seed=cin + cout + (k if isinstance(k, int) else 10 * k[0] + k[1]))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    wk = wd["c/kernel"].clone().requires_grad_(True)
    bk = wd["c/bias"].clone().requires_grad_(True)
    yr = K.conv2d(xr, wk, bk, stride, padding)
    dy = torch.randn(*yr.shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()

    xd = x.float().to(rt.device)
    y, ctx = layer.forward(xd)
    e_f = rel_err(y, yr)
    dx = layer.backward(ctx, dy.float().to(rt.device), True, True, 0)
    e_dx = rel_err(dx, xr.grad)
    e_dw = rel_err(ps.grad("c/kernel"), wk.grad)
    e_db = rel_err(ps.grad("c/bias"), bk.grad)
    report("conv2d cin=%d cout=%d k=%s s=%d pad=%s n=%d %dx%d  fwd=%.2e dx=%.2e dw=%.2e db=%.2e"
           % (cin, cout, k, stride, padding, n, h, w, e_f, e_dx, e_dw, e_db))
    assert e_f < TOL and e_dx < TOL and e_dw < TOL and e_db < TOL

    # residual-add epilogue of dgrad (used by the residual blocks' backward)
    res = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
    dx2 = layer.backward(ctx, dy.float().to(rt.device), True, False, 0, dx_residual=res.float().to(rt.device))
    assert rel_err(dx2, xr.grad + res) < TOL


@pytest.mark.parametrize("act", ["lrelu", "tanh"])
def test_conv2d_fused_activation(rt, act):
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    code = {"lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[act]
    layer = E.Conv2D("c", 16, 3 if act == "tanh" else 64, 3, 1, "same", code, 0.2)
    ps, wd = _standalone(rt, layer, seed=3)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 16, 10, 37, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    wk = wd["c/kernel"].clone().requires_grad_(True)
    z = K.conv2d(xr, wk, wd["c/bias"], 1, "same")
    yr = K.leaky_relu(z, 0.2) if act == "lrelu" else torch.tanh(z)
    dy = torch.randn(*yr.shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()
    y, ctx = layer.forward(x.float().to(rt.device))
    dx = layer.backward(ctx, dy.float().to(rt.device))
    e = (rel_err(y, yr), rel_err(dx, xr.grad), rel_err(ps.grad("c/kernel"), wk.grad))
    report("conv2d+%s fwd=%.2e dx=%.2e dw=%.2e" % ((act,) + e))
    assert max(e) < TOL


CONVT_CASES = [(64, 256, 3, 1, 8, 32), (64, 256, 3, 2, 7, 19), (256, 256, 3, 1, 6, 6), (64, 256, 5, 1, 9, 33),
               (16, 64, 3, 1, 4, 40),
               # the up-sampling path of make_generator_cyclegan: channels halve, down to 64 -> 64
               (64, 64, 3, 2, 24, 40), (128, 64, 3, 2, 12, 20), (256, 128, 3, 2, 6, 10)]


@pytest.mark.parametrize("cin,cout,k,n,h,w", CONVT_CASES)
def test_conv_transpose2d(rt, cin, cout, k, n, h, w):
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    layer = E.ConvT2D("t", cin, cout, k, L.ACT_LRELU, 0.2)
    ps, wd = _standalone(rt, layer, seed=k)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    wk = wd["t/kernel"].clone().requires_grad_(True)
    bk = wd["t/bias"].clone().requires_grad_(True)
    yr = K.leaky_relu(K.conv2d_transpose_same(xr, wk, bk, 2), 0.2)
    dy = torch.randn(*yr.shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()
    y, ctx = layer.forward(x.float().to(rt.device))
    dx = layer.backward(ctx, dy.float().to(rt.device))
    e = (rel_err(y, yr), rel_err(dx, xr.grad), rel_err(ps.grad("t/kernel"), wk.grad), rel_err(ps.grad("t/bias"), bk.grad))
    report("convT cin=%d cout=%d k=%d n=%d %dx%d fwd=%.2e dx=%.2e dw=%.2e db=%.2e" % ((cin, cout, k, n, h, w) + e))
    assert max(e) < TOL


NORM_CASES = [("batch", "prelu", 2, 64, 16, 32, True), ("batch", "none", 3, 64, 7, 9, True),
              ("instance", "prelu", 2, 256, 6, 10, True), ("instance", "prelu", 2, 256, 6, 10, False),      # make_generator_cyclegan's blocks at h/4
              ("batch", "lrelu", 2, 128, 8, 8, False), ("instance", "lrelu", 2, 128, 12, 12, False),
              ("batch", "lrelu", 8, 1024, 1, 1, False), ("batch", "prelu", 1, 64, 64, 64, True),
              # n*c > 65535 planes: Dense BatchNorm_1 (c = 1024) once D sees >= 64 frames (v1 wiring concatenates real + fake);
              # the backward kernels walk the planes beyond gridDim.y's limit
              ("batch", "lrelu", 72, 1024, 1, 1, False), ("batch", "prelu", 130, 512, 2, 2, False)]


@pytest.mark.parametrize("norm,act,n,c,h,w,residual", NORM_CASES)
def test_norm_act_fwd_bwd(rt, norm, act, n, c, h, w, residual):
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    code = {"prelu": L.ACT_PRELU, "lrelu": L.ACT_LRELU, "none": L.ACT_NONE}[act]
    layer = E.NormAct("bn", c, norm, code, 0.1, prelu_name="pr")
    ps, wd = _standalone(rt, layer, seed=c)
    g = torch.Generator().manual_seed(7)
    dense = (h == 1 and w == 1)
    shape = (n, c) if dense else (n, c, h, w)
    x = torch.randn(*shape, generator=g, dtype=torch.float64) * 1.7 + 0.4
    r = torch.randn(*shape, generator=g, dtype=torch.float64) if residual else None
    xr = x.clone().requires_grad_(True)
    leaf = {k: v.clone().requires_grad_(True) for k, v in wd.items()}

    def ref(training):
        if norm == "batch":
            z, mm, mv = K.batchnorm(xr, leaf["bn/gamma"], leaf["bn/beta"], leaf["bn/moving_mean"], leaf["bn/moving_variance"], training)
        else:
            z, mm, mv = K.instancenorm(xr), None, None
        if act == "prelu":
            z = K.prelu(z, leaf["pr/alpha"])
        elif act == "lrelu":
            z = K.leaky_relu(z, 0.1)
        return (z + r if residual else z), mm, mv

    xd = x.float().to(rt.device)
    rd = r.float().to(rt.device) if residual else None
    # inference mode (moving statistics)
    if norm == "batch":
        y0, _ = layer.forward(xd, False, residual=rd)
        assert rel_err(y0, ref(False)[0]) < TOL
    yr, mm, mv = ref(True)
    dy = torch.randn(*shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()
    y, ctx = layer.forward(xd, True, residual=rd)
    dx = layer.backward(ctx, dy.float().to(rt.device))
    errs = [rel_err(y, yr), rel_err(dx, xr.grad)]
    if norm == "batch":
        errs += [rel_err(ps.grad("bn/gamma"), leaf["bn/gamma"].grad), rel_err(ps.grad("bn/beta"), leaf["bn/beta"].grad),
                 rel_err(ps["bn/moving_mean"], mm), rel_err(ps["bn/moving_variance"], mv)]
    if act == "prelu":
        errs.append(rel_err(ps.grad("pr/alpha"), leaf["pr/alpha"].grad))
    report("norm=%s act=%s n=%d c=%d %dx%d res=%s errs=%s" % (norm, act, n, c, h, w, residual, ["%.1e" % e for e in errs]))
    assert max(errs) < TOL


def test_plain_prelu(rt):
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    layer = E.NormAct("op", 64, None, L.ACT_PRELU, prelu_name="pr")
    ps, wd = _standalone(rt, layer, seed=1)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 9, 20, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    al = wd["pr/alpha"].clone().requires_grad_(True)
    yr = K.prelu(xr, al)
    dy = torch.randn(*yr.shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()
    y, ctx = layer.forward(x.float().to(rt.device), True)
    dx = layer.backward(ctx, dy.float().to(rt.device))
    assert max(rel_err(y, yr), rel_err(dx, xr.grad), rel_err(ps.grad("pr/alpha"), al.grad)) < TOL


@pytest.mark.parametrize("b,cin,cout", [(8, 2048, 1024), (3, 1024, 32), (8, 32, 1), (2, 77, 130)])
def test_dense(rt, b, cin, cout):
    from upscaler import _engine as E
    from oracle import keras_ops as K
    layer = E.Dense("d", cin, cout)
    ps, wd = _standalone(rt, layer, seed=cout)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(b, cin, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    wk = wd["d/kernel"].clone().requires_grad_(True)
    bk = wd["d/bias"].clone().requires_grad_(True)
    yr = K.dense(xr, wk, bk)
    dy = torch.randn(*yr.shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()
    y, ctx = layer.forward(x.float().to(rt.device))
    dx = layer.backward(ctx, dy.float().to(rt.device))
    e = (rel_err(y, yr), rel_err(dx, xr.grad), rel_err(ps.grad("d/kernel"), wk.grad), rel_err(ps.grad("d/bias"), bk.grad))
    report("dense b=%d %d->%d fwd=%.2e dx=%.2e dw=%.2e db=%.2e" % ((b, cin, cout) + e))
    assert max(e) < TOL


def test_adam_losses_layout(rt):
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    lib = rt.lib
    g = torch.Generator().manual_seed(4)
    n = 100003
    p, gr, m, v = (torch.randn(n, generator=g, dtype=torch.float64) for _ in range(4))
    v = v.abs()
    pr, mr, vr = K.adam_keras_step(p, gr, m, v, 3)
    pd, gd, md, vd = (t.float().to(rt.device) for t in (p, gr, m, v))
    import math
    f32 = lambda v: float(np.float32(v))
    lr_t = f32(1e-3) * math.sqrt(1 - f32(0.999) ** 3) / (1 - f32(0.9) ** 3)          # upscaler.model.Adam.lr_t
    L.check(lib.vcg_adam_keras_multi(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, lr_t, 0.9, 0.999, 1e-7, 1.0, rt.stream), "adam")
    assert max(rel_err(pd, pr), rel_err(md, mr), rel_err(vd, vr)) < 1e-5
    # graph-replayable variant: step count on the device (t = *t_dev + 1), counter incremented by the call
    pd2, gd2, md2, vd2 = (t.float().to(rt.device) for t in (p, gr, m, v))
    t_dev = torch.tensor([2, 0], dtype=torch.int32, device=rt.device)          # {iteration count, lr_t scratch}
    L.check(lib.vcg_adam_keras_multi_dev(pd2.data_ptr(), gd2.data_ptr(), md2.data_ptr(), vd2.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-7,
                                         1.0, t_dev.data_ptr(), rt.stream), "adam_dev")
    assert int(t_dev[0].item()) == 3
    assert max(rel_err(pd2, pr), rel_err(md2, mr), rel_err(vd2, vr)) < 1e-5
    assert torch.equal(pd2, pd) and torch.equal(md2, md) and torch.equal(vd2, vd)      # lr_t on the device == the host's double
    # grad_scale: the 1/ranks of a data-parallel SUM bucket folded into the update (g*4 with scale 0.25 == g: exact in binary)
    pd3, gd3, md3, vd3 = (t.float().to(rt.device) for t in (p, 4 * gr, m, v))
    L.check(lib.vcg_adam_keras_multi(pd3.data_ptr(), gd3.data_ptr(), md3.data_ptr(), vd3.data_ptr(), n, lr_t, 0.9, 0.999, 1e-7, 0.25, rt.stream), "adam")
    assert torch.equal(pd3, pd) and torch.equal(md3, md) and torch.equal(vd3, vd)
    # pixel loss
    a, b = torch.randn(2, 3, 17, 19, generator=g, dtype=torch.float64), torch.randn(2, 3, 17, 19, generator=g, dtype=torch.float64)
    for kind, code in (("mse", L.LOSS_MSE), ("mae", L.LOSS_MAE)):
        ar = a.clone().requires_grad_(True)
        lv = ((ar - b) ** 2).mean() if kind == "mse" else (ar - b).abs().mean()
        (0.7 * lv).backward()
        out, da = rt.empty(1), rt.empty(*a.shape)
        ws, wsn = rt.workspace(lib.vcg_mean_reduce_workspace_bytes(a.numel()))
        a_dev, b_dev = a.float().to(rt.device), b.float().to(rt.device)      # keep the buffers alive
        L.check(lib.vcg_pixel_loss(a_dev.data_ptr(), b_dev.data_ptr(), a.numel(), code, 0.7,
                                   out.data_ptr(), da.data_ptr(), ws, wsn, rt.stream), "pixel_loss")
        assert abs(out.item() - lv.item()) < 1e-5 * abs(lv.item()) + 1e-7
        assert rel_err(da, ar.grad) < 1e-5
    # layout + uint8 edge
    from upscaler import data as D
    from oracle import data as OD
    u8 = torch.randint(0, 256, (2, 9, 11, 3), generator=g, dtype=torch.uint8)
    dev = D.frames_u8_to_device(u8)
    ref = torch.tensor(OD.convert_uint8_to_array(u8.numpy())).permute(0, 3, 1, 2).float()
    assert torch.equal(dev.cpu(), ref)
    back = D.device_to_frames_u8(dev).cpu()
    assert torch.equal(back, u8)
    xs = torch.rand(3, 8, 8, 3, generator=g) * 2 - 1
    q = D.device_to_frames_u8(E.to_device_nchw(rt, xs)).cpu().numpy()
    assert np.array_equal(q, OD.convert_array_to_uint8(xs.numpy().astype(np.float32)))
    assert torch.equal(E.to_nhwc(rt, E.to_device_nchw(rt, xs)).cpu(), xs)
    m = E.mean_scalar(rt, dev)
    assert abs(m.item() - ref.double().mean().item()) < 1e-6


def test_assemble_training_batch_matches_loop_body(rt):
    """train_gan3.py:341-345 on the device: same arrays as pd.concat + convert_image_series_to_array"""
    from upscaler import data as PD
    rng = np.random.RandomState(2)
    hd = rng.randint(0, 256, (3, 16, 24, 3)).astype(np.uint8)
    g1, g2, sc = (rng.randint(0, 256, (3, 8, 12, 3)).astype(np.uint8) for _ in range(3))
    lr, hr = PD.assemble_training_batch(hd, [f for f in g1], g2, torch.from_numpy(sc))
    ref_hr = PD.convert_image_series_to_array(list(hd) * 3)
    ref_lr = PD.convert_image_series_to_array(list(g1) + list(g2) + list(sc))
    assert np.array_equal(lr.permute(0, 2, 3, 1).cpu().numpy(), ref_lr.astype(np.float32))
    assert np.array_equal(hr.permute(0, 2, 3, 1).cpu().numpy(), ref_hr.astype(np.float32))


def test_final_conv_rowchain_tanh(rt):
    """final/conv with its fused tanh at a width the row-chain kernel serves (256 -> 3, 9x9, w % 64 == 0)"""
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    layer = E.Conv2D("c", 256, 3, 9, 1, "same", L.ACT_TANH)
    ps, wd = _standalone(rt, layer, seed=5)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 37, 128, generator=g, dtype=torch.float64) * 0.3
    yr = torch.tanh(K.conv2d(x, wd["c/kernel"], wd["c/bias"], 1, "same"))
    y, _ = layer.forward(x.float().to(rt.device))
    e = rel_err(y, yr)
    report("final conv (row-chain kernel) + tanh 256->3 n=2 37x128 fwd=%.2e" % e)
    assert e < TOL


HEADS = ["none", "sigmoid", "log-sigm", "tanh", "bi-log"]


@pytest.mark.parametrize("kind", HEADS)
def test_head_activation_fwd_bwd(rt, kind):
    """discriminator output activations (model.py:885-892): value and derivative against autograd of the oracle's
    expression, on inputs that include 0, large magnitudes of both signs and the unbounded critic outputs the reference
    was observed to produce"""
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    g = torch.Generator().manual_seed(5)
    z = torch.cat([torch.randn(997, generator=g, dtype=torch.float64) * 3, torch.tensor([0.0, 1e-6, -1e-6, 20.0, -20.0, 60.0, -60.0, 1e4, -1e4])])
    if kind == "log-sigm":
        z = z.clamp(min=-700.0)           # the fp64 oracle's naive log(sigmoid) is finite down to here
    z = z.float().double().view(-1, 1)
    zr = z.clone().requires_grad_(True)
    yr = K.head_activation(zr, kind)
    dy = torch.randn(*z.shape, generator=g, dtype=torch.float64)
    (yr * dy).sum().backward()
    zd = z.float().to(rt.device)
    y = E.head_act_fwd(rt, zd, L.HEAD_KINDS[kind])
    dz = E.head_act_bwd(rt, zd, dy.float().to(rt.device), L.HEAD_KINDS[kind])
    # element-wise relative error (values span many orders of magnitude)
    ev = float(((y.cpu().double() - yr.detach()).abs() / (yr.detach().abs() + 1e-6)).max())
    ed = float(((dz.cpu().double() - zr.grad).abs() / (zr.grad.abs() + 1e-6)).max())
    report("head activation %-8s value err=%.2e derivative err=%.2e" % (kind, ev, ed))
    assert ev < 1e-5 and ed < 1e-5


@pytest.mark.parametrize("loss_act", HEADS)
@pytest.mark.parametrize("head", HEADS)
def test_gan_loss_all_activation_pairs(rt, head, loss_act):
    """D head activation x GanLosses.loss_activation, all 25 pairs (train_gan3.py:58,63): the relativistic discriminator
    and generator losses act(mean(D_a) - mean(D_b)) and their gradients wrt the critic's PRE-activation outputs, evaluated
    entirely on the device (vcg_head_act_* + vcg_mean_reduce + vcg_gan_loss), against autograd of model.py:244-259."""
    from upscaler import _engine as E, _lib as L
    from oracle import keras_ops as K
    g = torch.Generator().manual_seed(HEADS.index(head) * 7 + HEADS.index(loss_act))
    za = (torch.randn(6, 1, generator=g, dtype=torch.float64) * 2 + 0.3).float().double()
    zb = (torch.randn(6, 1, generator=g, dtype=torch.float64) * 2 - 0.2).float().double()
    zar, zbr = za.clone().requires_grad_(True), zb.clone().requires_grad_(True)
    loss = K.head_activation(K.head_activation(zar, head).mean() - K.head_activation(zbr, head).mean(), loss_act)
    loss.backward()
    hk, lk = L.HEAD_KINDS[head], L.HEAD_KINDS[loss_act]
    zad, zbd = za.float().to(rt.device), zb.float().to(rt.device)
    means, out = rt.zeros(2), rt.zeros(1)
    E.mean_scalar(rt, E.head_act_fwd(rt, zad, hk), out=means[0:1])
    E.mean_scalar(rt, E.head_act_fwd(rt, zbd, hk), out=means[1:2])
    da, db = rt.empty(6, 1), rt.empty(6, 1)
    E.gan_loss(rt, means[0:1], means[1:2], 1.0, lk, out, da, 1.0 / 6, db, -1.0 / 6)
    dza, dzb = E.head_act_bwd(rt, zad, da, hk), E.head_act_bwd(rt, zbd, db, hk)
    ev = abs(float(out.item()) - float(loss)) / (abs(float(loss)) + 1e-6)
    eg = max(rel_err(dza, zar.grad), rel_err(dzb, zbr.grad))
    report("gan loss head=%-8s loss_activation=%-8s value err=%.2e grad err=%.2e" % (head, loss_act, ev, eg))
    assert ev < 1e-5 and eg < 1e-5
    # summed means of 2 "ranks" with mean_scale 1/2 (the data-parallel form) give the same loss
    means2 = rt.zeros(2)
    E.axpby(rt, means, means2, 2.0, 0.0)
    out2 = rt.zeros(1)
    E.gan_loss(rt, means2[0:1], means2[1:2], 0.5, lk, out2, da, 1.0 / 6, db, -1.0 / 6)
    assert torch.equal(out2, out)


# ---- resize / crop / pad / channel copies / dropout (generators behind train_gan3.py -gm: model.py:332-363, 505-827) ----------------
@pytest.mark.parametrize("factor", [2, 3, 4])
@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
def test_resize2d_is_tf1_resize_images(rt, factor, mode):
    from oracle import models as M
    from upscaler import _lib as L
    x = torch.randn(2, 3, 13, 17)
    xd = x.to(rt.device)
    y = rt.empty(2, 3, 13 * factor, 17 * factor)
    L.check(rt.lib.vcg_resize2d(xd.data_ptr(), y.data_ptr(), 6, 13, 17, factor, 1 if mode == "bilinear" else 0, rt.stream), "vcg_resize2d")
    ref = M.resize_images_tf1(x.double(), factor, mode)
    if mode == "nearest":
        assert torch.equal(y.cpu().double(), ref)
    else:
        # the kernel forms the source coordinate as TF does, y * (in / (float)out) in fp32: exact for the power-of-two factors the
        # reference uses, one fp32 rounding of the lerp weight (times the coordinate) for factor 3
        assert float((y.cpu().double() - ref).abs().max()) < (2e-6 if factor != 3 else 1e-5)


def test_crop_pad_and_channel_copies(rt):
    from upscaler import _lib as L
    x = torch.randn(3, 5, 11, 14).to(rt.device)
    y = rt.empty(3, 5, 8, 9)
    L.check(rt.lib.vcg_crop2d(x.data_ptr(), y.data_ptr(), 15, 11, 14, 1, 3, 8, 9, rt.stream), "vcg_crop2d")
    assert torch.equal(y, x[:, :, 1:9, 3:12])
    back = rt.empty(3, 5, 11, 14)
    L.check(rt.lib.vcg_pad2d(y.data_ptr(), back.data_ptr(), 15, 8, 9, 1, 3, 11, 14, rt.stream), "vcg_pad2d")
    ref = torch.zeros_like(x)
    ref[:, :, 1:9, 3:12] = x[:, :, 1:9, 3:12]
    assert torch.equal(back, ref)
    assert rt.lib.vcg_crop2d(x.data_ptr(), y.data_ptr(), 15, 11, 14, 4, 3, 8, 9, rt.stream) == -2       # VCG_E_SHAPE: the window leaves the source
    a, b = torch.randn(2, 3, 6, 7).to(rt.device), torch.randn(2, 224, 6, 7).to(rt.device)
    cat = rt.empty(2, 227, 6, 7)
    L.check(rt.lib.vcg_copy_channels(a.data_ptr(), cat.data_ptr(), 2, 3, 0, 227, 0, 3, 42, rt.stream), "copy a")
    L.check(rt.lib.vcg_copy_channels(b.data_ptr(), cat.data_ptr(), 2, 224, 0, 227, 3, 224, 42, rt.stream), "copy b")
    assert torch.equal(cat, torch.cat([a, b], 1))
    part = rt.empty(2, 100, 6, 7)
    L.check(rt.lib.vcg_copy_channels(cat.data_ptr(), part.data_ptr(), 2, 227, 50, 100, 0, 100, 42, rt.stream), "slice")
    assert torch.equal(part, cat[:, 50:150])


def test_dropout_is_tf_nn_dropout_given_the_mask(rt):
    from upscaler import _lib as L
    n = 1 << 20
    x = torch.randn(n).to(rt.device)
    step = torch.zeros(1, dtype=torch.int64, device=rt.device)
    outs = []
    for it in range(2):
        y, mask = rt.empty(n), torch.empty(n, dtype=torch.uint8, device=rt.device)
        L.check(rt.lib.vcg_dropout_fwd(x.data_ptr(), y.data_ptr(), mask.data_ptr(), n, 0.1, 1234, step.data_ptr(), rt.stream), "vcg_dropout_fwd")
        keep = torch.tensor(0.9, dtype=torch.float32, device=rt.device)          # tf.nn.dropout: div(x, keep_prob) * mask, a true division
        assert torch.equal(y, torch.where(mask.bool(), torch.div(x, keep), torch.zeros_like(x)))
        assert abs(float(mask.float().mean()) - 0.9) < 2e-3                   # keep probability 1 - rate (sigma = 3e-4)
        dy, dx = torch.randn(n).to(rt.device), rt.empty(n)
        L.check(rt.lib.vcg_dropout_bwd(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), n, 0.1, rt.stream), "vcg_dropout_bwd")
        assert torch.equal(dx, torch.where(mask.bool(), torch.div(dy, keep), torch.zeros_like(dy)))
        outs.append(mask.clone())
        L.check(rt.lib.vcg_counter_inc(step.data_ptr(), rt.stream), "vcg_counter_inc")
    assert int(step.item()) == 2
    both = float((outs[0] & outs[1]).float().mean())
    assert abs(both - 0.81) < 3e-3                                            # the two steps' masks are independent


STATS_EPILOGUE_CASES = [
    # cin, cout, k, stride, padding, n, h, w, instance
    (64, 64, 3, 1, "same", 2, 64, 128, False),       # the generator trunk's layer on 64-column tiles (two MFMA x-tiles per wave)
    (64, 64, 3, 1, "same", 3, 37, 45, False),        # ragged rows and columns: the pixels past the image must not count
    (64, 64, 3, 1, "same", 2, 19, 23, True),         # instance norm: records per image, 32-column tiles
    (64, 128, 4, 2, 1, 2, 66, 50, False),            # PatchGAN block 2 (4x4 stride 2), two output-channel blocks
    (128, 256, 4, 1, 1, 2, 20, 21, False),           # PatchGAN block 4's kernel (4x4 stride 1), large bias against a small spread
    (64, 96, 3, 2, "same", 2, 31, 33, True),         # a ragged channel block (96 = 64 + 32), stride 2
]


@pytest.mark.parametrize("cin,cout,k,stride,padding,n,h,w,instance", STATS_EPILOGUE_CASES)
def test_conv2d_stats_epilogue_matches_the_statistics_pass(rt, cin, cout, k, stride, padding, n, h, w, instance):
    """vcg_conv2d_fwd_stats + vcg_norm_finalize_partials_shifted (the statistics of the normalisation behind a convolution from its epilogue)
    against the fp64 statistics of the layer's output, and NormAct.forward(stats=...) against the separate statistics pass: same output, same
    moving averages."""
    from oracle import keras_ops as K
    from upscaler import _engine as E, _lib as L
    conv = E.Conv2D("c", cin, cout, k, stride, padding)
    ps, w64 = _standalone(rt, conv, seed=cin + cout + h)
    ps.set_weights({"c/bias": (np.random.RandomState(h).uniform(-3.0, 3.0, cout)).astype(np.float32)})       # |mean| >> spread for some channels
    norm = E.NormAct("nm", cout, "instance" if instance else "batch", L.ACT_PRELU, 0.0, prelu_name="pr")
    norm2 = E.NormAct("nm", cout, "instance" if instance else "batch", L.ACT_PRELU, 0.0, prelu_name="pr")
    psn, _ = _standalone(rt, norm, seed=3)
    psn2, _ = _standalone(rt, norm2, seed=3)
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(n, cin, h, w, generator=g).to(rt.device)
    y, _, st = conv.forward_stats(x, instance)
    assert st is not None, "this shape must be served by the statistics epilogue"
    y_ref, _ = conv.forward(x)
    assert torch.equal(y, y_ref)
    out, ctx = norm.forward(y, True, stats=st)
    out2, ctx2 = norm2.forward(y_ref, True)
    yd = y_ref.double().cpu()
    dims = (2, 3) if instance else (0, 2, 3)
    mean64, var64 = yd.mean(dims), yd.var(dims, unbiased=False)
    mean, invstd = ctx[1]
    e_mean = float(((mean.cpu().double().reshape(mean64.shape) - mean64).abs() / (var64.sqrt() + 1e-6)).max())
    e_is = rel_err(invstd.cpu().double().reshape(var64.shape), 1.0 / torch.sqrt(var64 + (1e-5 if instance else 1e-3)))
    e_out = rel_err(out, out2)
    report("fp32 conv stats epilogue %d->%d k%d s%d n=%d %dx%d %s: mean (in sigmas)=%.2e invstd=%.2e out vs stats pass=%.2e"
           % (cin, cout, k, stride, n, h, w, "instance" if instance else "batch", e_mean, e_is, e_out))
    assert e_mean < 1e-5 and e_is < 1e-5 and e_out < 1e-5
    if not instance:
        for nm in ("nm/moving_mean", "nm/moving_variance"):
            assert rel_err(psn[nm], psn2[nm]) < 1e-5
