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
