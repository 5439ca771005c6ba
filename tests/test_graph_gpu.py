"""The reference's block functions (residual_block / upsampling_block / downsampling_block, model.py:15-27,63-75) on
the functional graph API: (1) make_upscaler_orig written block by block reproduces the hand-wired generator (forward bit for bit);
(2) a custom generator using the (reference-unused) downsampling_block matches an oracle restatement."""
import numpy as np
import pytest
import torch

from conftest import rel_err, report

pytestmark = pytest.mark.gpu


def _frames(seed, n, h, w):
    return (np.random.RandomState(seed).randint(0, 256, (n, h, w, 3)) / 127.5 - 1).astype(np.float32)


def test_functional_upscaler_is_bit_identical_to_hand_wired(rt):
    from upscaler import _engine as E, _lib as L, model as PM
    A = PM.make_upscaler_orig((64, 64, 3), kernel_size=3, upscale_factor=2, res_block_num=2)
    B = PM.make_upscaler_orig_functional((64, 64, 3), kernel_size=3, upscale_factor=2, res_block_num=2)
    wa, wb = A.get_weights_dict(), B.get_weights_dict()
    assert list(wa) == list(wb)                                   # same layer / weight names, same order
    assert all(np.array_equal(wa[k], wb[k]) for k in wa)          # same seeded Glorot init
    assert A.output_shape == B.output_shape == (None, 64, 64, 3)
    x, t = _frames(1, 2, 32, 32), _frames(2, 2, 64, 64)
    assert np.array_equal(A.predict(x), B.predict(x))
    outs = []
    for m in (A, B):
        y, tape = m.forward(E.to_device_nchw(rt, x), True)
        val, dy = rt.empty(1), rt.empty(*y.shape)
        ws, wsn = rt.workspace(4096)
        td = E.to_device_nchw(rt, t)
        L.check(rt.lib.vcg_pixel_loss(y.data_ptr(), td.data_ptr(), y.numel(), L.LOSS_MSE, 1.0, val.data_ptr(), dy.data_ptr(), ws, wsn,
                                      rt.stream), "pixel_loss")
        m.backward(tape, dy, 0)
        outs.append((y.cpu(), m.ps.grads.cpu().clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    # gradients: same kernels, but the skip-connection gradients are summed in a different order
    ga, gb = outs[0][1], outs[1][1]
    assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())


def test_custom_generator_with_downsampling_block_matches_oracle(rt):
    """Input -> Conv 3x3 -> downsampling_block(s2) -> residual_block -> upsampling_block -> Conv 3x3 + tanh"""
    from oracle import keras_ops as K
    from upscaler import _engine as E, _lib as L, model as PM
    inp = PM.Input((32, 48, 3))
    m = PM.conv2d(inp, 64, 3, 1, name="stem")
    m = PM.downsampling_block(m, 3, 64, 2, name="down")
    m = PM.residual_block(m, 3, 64, 1, name="rb")
    m = PM.upsampling_block(m, 3, 64, 2, name="up")
    m = PM.conv2d(m, 3, 3, 1, activation="tanh", name="head")
    net = PM.build_model(inp, m, seed=3)
    assert net.output_shape == (None, 32, 48, 3)
    w = net.get_weights_dict()
    rng = np.random.RandomState(0)
    for k in w:                                                   # non-trivial norm / slope parameters
        if k.endswith(("/beta", "/bias")):
            w[k] = rng.uniform(-0.1, 0.1, w[k].shape).astype(np.float32)
        if k.endswith("/alpha"):
            w[k] = rng.uniform(0, 0.3, w[k].shape).astype(np.float32)
    net.set_weights_dict(w)
    x, t = _frames(5, 2, 32, 48), _frames(6, 2, 32, 48)
    leaf = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(("moving_mean", "moving_variance"))) for k, v in w.items()}

    def ref(xn):
        h = K.conv2d(xn, leaf["stem/kernel"], leaf["stem/bias"], 1, "same")
        h = K.leaky_relu(K.conv2d(h, leaf["down/kernel"], leaf["down/bias"], 2, "same"), 0.2)
        g = h
        h = K.conv2d(h, leaf["rb/conv_pre/kernel"], leaf["rb/conv_pre/bias"], 1, "same")
        h, _, _ = K.batchnorm(h, leaf["rb/batch_norm_pre/gamma"], leaf["rb/batch_norm_pre/beta"], leaf["rb/batch_norm_pre/moving_mean"],
                              leaf["rb/batch_norm_pre/moving_variance"], True)
        h = K.prelu(h, leaf["rb/prelu/alpha"])
        h = K.conv2d(h, leaf["rb/conv_post/kernel"], leaf["rb/conv_post/bias"], 1, "same")
        h, _, _ = K.batchnorm(h, leaf["rb/batch_norm_post/gamma"], leaf["rb/batch_norm_post/beta"], leaf["rb/batch_norm_post/moving_mean"],
                              leaf["rb/batch_norm_post/moving_variance"], True)
        h = g + h
        h = K.leaky_relu(K.conv2d_transpose_same(h, leaf["up/conv_transp/kernel"], leaf["up/conv_transp/bias"], 2), 0.2)
        return torch.tanh(K.conv2d(h, leaf["head/kernel"], leaf["head/bias"], 1, "same"))

    yr = ref(torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2))
    loss = ((yr - torch.tensor(t, dtype=torch.float64).permute(0, 3, 1, 2)) ** 2).mean()
    names = [k for k, v in leaf.items() if v.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(loss, [leaf[k] for k in names])))
    y, tape = net.forward(E.to_device_nchw(rt, x), True)
    e_f = rel_err(y, yr)
    val, dy = rt.empty(1), rt.empty(*y.shape)
    ws, wsn = rt.workspace(4096)
    td = E.to_device_nchw(rt, t)
    L.check(rt.lib.vcg_pixel_loss(y.data_ptr(), td.data_ptr(), y.numel(), L.LOSS_MSE, 1.0, val.data_ptr(), dy.data_ptr(), ws, wsn, rt.stream),
            "pixel_loss")
    net.backward(tape, dy, 0)
    gmax = max(float(g.abs().max()) for g in grads.values())
    worst = max(float((net.ps.grad(k).cpu().double() - grads[k]).abs().max() / (grads[k].abs().max() + 1e-3 * gmax)) for k in names)
    report("custom generator (downsampling_block) fwd err=%.2e worst grad err=%.2e" % (e_f, worst))
    assert e_f < 1e-3 and worst < 5e-3


@pytest.mark.parametrize("factor", [2, 4])
def test_upscaler_attention_matches_oracle(rt, factor):
    """make_upscaler_attention (model.py:299-328), train_gan3.py's default generator, at upscale_factor 2 (BASELINE.json's) and 4
    (train_gan3.py's default -d; the second up-sampling block resizes the input x2 and transposes it with strides 4): parameter
    count, inference and training-mode forward, every parameter gradient against the fp64 oracle (autograd)."""
    from oracle import models as M
    from upscaler import _engine as E, model as PM
    out_shape, res, k = (64, 96, 3), 3, 3
    gw = M.init_upscaler_attention(out_shape, k, 64, factor, res, seed=5)
    rng = np.random.RandomState(2)
    for n_, v in gw.items():                              # non-trivial BN / PReLU parameters
        if n_.endswith(("/bias", "/beta", "/moving_mean")):
            gw[n_] = rng.uniform(-0.1, 0.1, v.shape).astype(np.float32)
        elif n_.endswith(("/gamma", "/moving_variance")):
            gw[n_] = rng.uniform(0.8, 1.2, v.shape).astype(np.float32)
        elif n_.endswith("/alpha"):
            gw[n_] = rng.uniform(0.0, 0.3, v.shape).astype(np.float32)
    G = PM.make_upscaler_attention(out_shape, kernel_size=k, upscale_factor=factor, res_block_num=res)
    assert G.count_params() == M.count_params(gw)
    assert G.input_shape == (None, 64 // factor, 96 // factor, 3) and G.output_shape == (None, 64, 96, 3)
    G.set_weights_dict(gw)
    x = (rng.randint(0, 256, (2, 64 // factor, 96 // factor, 3)) / 127.5 - 1).astype(np.float32)
    t = (rng.randint(0, 256, (2, 64, 96, 3)) / 127.5 - 1).astype(np.float32)
    with torch.no_grad():
        y0, _ = M.upscaler_attention_forward(M.to_torch(gw, torch.float64), torch.tensor(x, dtype=torch.float64), False, res, factor)
    e0 = rel_err(torch.tensor(G.predict(x)), y0)
    leaf = M.to_torch(gw, torch.float64, requires_grad=True)
    y, upd = M.upscaler_attention_forward(leaf, torch.tensor(x, dtype=torch.float64), True, res, factor)
    loss = ((y - torch.tensor(t, dtype=torch.float64)) ** 2).mean()
    names = [n_ for n_, v in leaf.items() if v.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(loss, [leaf[n_] for n_ in names])))
    leaf32 = M.to_torch(gw, torch.float32, requires_grad=True)
    y32, _ = M.upscaler_attention_forward(leaf32, torch.tensor(x), True, res, factor)
    g32 = dict(zip(names, torch.autograd.grad(((y32 - torch.tensor(t)) ** 2).mean(), [leaf32[n_] for n_ in names])))
    yd, tape = G.forward(E.to_device_nchw(rt, x), True)
    e1 = rel_err(E.to_nhwc(rt, yd), y)
    val, dy = PM._pixel_loss(rt, yd, E.to_device_nchw(rt, t), "mse", 1.0)
    G.backward(tape, dy, 0)
    gmax = max(float(g.abs().max()) for g in grads.values())
    worst = 0.0
    for n_ in names:
        a, b = G.ps.grad(n_).cpu().double(), grads[n_]
        err = float((a - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
        e32 = float((g32[n_].double() - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
        worst = max(worst, err)
        assert err < max(1e-3, 4 * e32), (n_, err, e32)
    sw = G.get_weights_dict()
    for n_, v in upd.items():
        assert np.max(np.abs(sw[n_] - v.detach().numpy())) < 1e-4 * (np.max(np.abs(v.detach().numpy())) + 1e-3), n_
    report("make_upscaler_attention x%d: predict err=%.2e train fwd err=%.2e worst gradient err=%.2e" % (factor, e0, e1, worst))
    assert e0 < 1e-3 and e1 < 1e-3


def test_train_gan3_default_model_choice_trains(rt):
    """train_gan3.py's default model and loss selection (:55-63: -gm resnet-att, -dm s512, -da bi-log, -dl rel, -dla log-sigm)
    through make_and_compile_gan2 and two loop-body iterations, against the fp64 oracle (x2, pixel-MSE content loss)."""
    from oracle import models as M, train as T
    from upscaler import model as PM, _lib as L
    res, k, bs = 2, 3, 4
    gw = M.init_upscaler_attention((128, 128, 3), k, 64, 2, res, seed=7)
    dw = M.init_discriminator_512((128, 128, 3), "simple", seed=11)
    G = PM.make_upscaler_attention((128, 128, 3), kernel_size=k, upscale_factor=2, res_block_num=res)
    D = PM.make_discriminator_simple_512((128, 128, 3), activation="bi-log")
    G.set_weights_dict(gw)
    D.set_weights_dict(dw)
    gen_train, disc_train, gan_train = PM.make_and_compile_gan2(
        G, D, (64, 64, 3), (128, 128, 3), "mse", 1, lambda: PM.RelativisticLosses(loss_activation="log-sigm"), 1e-2, optimizer=PM.Adam())
    tr = gan_train.trainer
    for s in (tr.g_slots, tr.d_slots):
        L.check(rt.lib.vcg_fill(s.v.data_ptr(), s.v.numel(), 1.0, rt.stream), "vcg_fill")
    orc = T.GanOracle(lambda w, x, t: M.upscaler_attention_forward(w, x, t, res, 2), M.to_torch(gw, torch.float64),
                      lambda w, x, t: M.discriminator_512_forward(w, x, t, activation="bi-log"), M.to_torch(dw, torch.float64),
                      losses="rel", loss_activation="log-sigm", discriminator_loss_weight=1e-2, adam_v0=1.0)
    for it in range(2):
        lr, hr = _frames(50 + it, bs, 64, 64), _frames(60 + it, bs, 128, 128)
        fake = gen_train.predict(lr)
        got = (disc_train.train_on_batch([hr, fake], -np.ones(bs)),) + tuple(gan_train.train_on_batch([lr, hr], [hr, -np.ones(bs)]))
        ref = orc.train_step(torch.tensor(lr, dtype=torch.float64), torch.tensor(hr, dtype=torch.float64))
        scale = max(abs(v) for v in ref)
        report("train_gan3 default models it=%d got=%s ref=%s" % (it, ["%.6g" % v for v in got], ["%.6g" % v for v in ref]))
        for a, b in zip(got, ref):
            assert abs(a - b) < 1e-3 * scale, (got, ref)
