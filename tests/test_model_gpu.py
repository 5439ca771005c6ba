"""End-to-end GPU parity of the drop-in API (upscaler.model) against the CPU oracle: generator and
discriminator forward, and the reference's three-call train step (train_gan3.py:346-354) for two
iterations -- losses, updated weights and BN moving statistics."""
import numpy as np
import pytest
import torch

from conftest import rel_err, report

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _frames(seed, n, h, w):
    rng = np.random.RandomState(seed)
    return (rng.randint(0, 256, (n, h, w, 3)) / 127.5 - 1).astype(np.float32)


def _perturb(w, seed):
    """make BN/PReLU parameters non-trivial so that every backward path is exercised"""
    rng = np.random.RandomState(seed)
    out = {}
    for k, v in w.items():
        v = np.array(v, np.float32)
        if k.endswith(("/bias", "/beta")):
            v = rng.uniform(-0.1, 0.1, v.shape).astype(np.float32)
        elif k.endswith("/gamma"):
            v = rng.uniform(0.8, 1.2, v.shape).astype(np.float32)
        elif k.endswith("/alpha"):
            v = rng.uniform(0.0, 0.3, v.shape).astype(np.float32)
        elif k.endswith("/moving_mean"):
            v = rng.uniform(-0.1, 0.1, v.shape).astype(np.float32)
        elif k.endswith("/moving_variance"):
            v = rng.uniform(0.8, 1.2, v.shape).astype(np.float32)
        out[k] = v
    return out


def _oracle_pair(kernel_size, res, disc, dtype=torch.float64):
    from oracle import models as M
    gw = _perturb(M.init_upscaler_orig((128, 128, 3), kernel_size, 64, 2, res, seed=7), 1)
    if disc == "patch":
        dw = _perturb(M.init_discriminator_patchgan_70((128, 128, 3), seed=11), 2)
        df = lambda w, x, t: M.discriminator_patchgan_70_forward(w, x, t)
    else:
        dw = _perturb(M.init_discriminator_512((128, 128, 3), disc, seed=11), 2)
        df = lambda w, x, t: M.discriminator_512_forward(w, x, t)
    gf = lambda w, x, t: M.upscaler_orig_forward(w, x, t, res, 2)
    return gw, dw, gf, df


def _product_pair(kernel_size, res, disc, gw, dw):
    from upscaler import model as PM
    G = PM.make_upscaler_orig((128, 128, 3), kernel_size=kernel_size, upscale_factor=2, res_block_num=res)
    if disc == "patch":
        D = PM.make_discriminator_patchgan_70((128, 128, 3))
    elif disc == "simple":
        D = PM.make_discriminator_simple_512((128, 128, 3))
    else:
        D = PM.make_discriminator_thin_512((128, 128, 3))
    G.set_weights_dict(gw)
    D.set_weights_dict(dw)
    return G, D


@pytest.mark.parametrize("k", [3, 5])
def test_generator_forward(rt, k):
    from oracle import models as M
    from upscaler import _engine as E
    gw, dw, gf, df = _oracle_pair(k, 2, "simple")
    G, D = _product_pair(k, 2, "simple", gw, dw)
    assert G.count_params() == M.count_params(gw)
    x = _frames(3, 2, 64, 64)
    wt = M.to_torch(gw, torch.float64)
    for training in (False, True):
        with torch.no_grad():
            yr, _ = gf(wt, torch.tensor(x, dtype=torch.float64), training)
        if training:
            y, _ = G.forward(E.to_device_nchw(rt, x), True)
            y = E.to_nhwc(rt, y)
        else:
            y = torch.tensor(G.predict(x))
        e = rel_err(y, yr)
        report("generator k=%d training=%s fwd err=%.2e" % (k, training, e))
        assert e < TOL
    assert G.predict(x).shape == (2, 128, 128, 3)
    # shape-polymorphic inference (upscaler_mini_testing.ipynb cells 5-7)
    x2 = _frames(4, 1, 40, 72)
    with torch.no_grad():
        yr, _ = gf(wt, torch.tensor(x2, dtype=torch.float64), False)
    assert rel_err(torch.tensor(G.predict(x2)), yr) < TOL


@pytest.mark.parametrize("disc", ["simple", "thin", "patch"])
def test_discriminator_forward(rt, disc):
    from oracle import models as M
    gw, dw, gf, df = _oracle_pair(3, 1, disc)
    G, D = _product_pair(3, 1, disc, gw, dw)
    assert D.count_params() == M.count_params(dw)
    x = _frames(5, 3, 128, 128)
    with torch.no_grad():
        yr, _ = df(M.to_torch(dw, torch.float64), torch.tensor(x, dtype=torch.float64), False if disc != "patch" else True)
    y = torch.tensor(D.predict(x))
    e = rel_err(y, yr)
    report("discriminator %s predict err=%.2e out=%s" % (disc, e, tuple(y.shape)))
    assert e < TOL


CASES = [("gan2", "wass", "simple", 3), ("gan2", "rel", "thin", 3), ("v1", "wass", "simple", 3), ("gan2", "wass", "patch", 3),
         ("gan2", "wass", "simple", 5)]


@pytest.mark.parametrize("wiring,losses,disc,k", CASES)
def test_train_step_parity(rt, wiring, losses, disc, k):
    """two loop-body iterations; compares the four reported losses, every updated weight of G and D
    and the BN moving statistics with the fp64 oracle."""
    from oracle import models as M, train as T
    from upscaler import model as PM
    res = 2
    gw, dw, gf, df = _oracle_pair(k, res, disc)
    G, D = _product_pair(k, res, disc, gw, dw)
    orc = T.GanOracle(gf, M.to_torch(gw, torch.float64), df, M.to_torch(dw, torch.float64), wiring=wiring, content="mse",
                      content_loss_weight=1.0, losses=losses, loss_activation="log-sigm", discriminator_loss_weight=1e-2)
    opt = PM.Adam()
    if wiring == "gan2":
        fac = (lambda: PM.WassersteinLosses()) if losses == "wass" else (lambda: PM.RelativisticLosses(loss_activation="log-sigm"))
        gen_train, disc_train, gan_train = PM.make_and_compile_gan2(G, D, (64, 64, 3), (128, 128, 3), "mse", 1.0, fac, 1e-2, optimizer=opt)
    else:
        gen_train, disc_train, gan_train = PM.make_and_compile_gan(G, D, (64, 64, 3), (128, 128, 3), "mse", 1.0, PM.wasserstein_loss, 1e-2, optimizer=opt)
    bs = 2
    for it in range(2):
        lr, hr = _frames(10 + it, bs, 64, 64), _frames(20 + it, bs, 128, 128)
        # reference loop body, train_gan3.py:346-354 / train_gan.py:303-317
        fake = gen_train.predict(lr)
        if wiring == "gan2":
            loss_disc = disc_train.train_on_batch([hr, fake], -np.ones(bs))
            loss_gan = gan_train.train_on_batch([lr, hr], [hr, -np.ones(bs)])
        else:
            loss_disc = disc_train.train_on_batch(np.concatenate((hr, fake), 0), np.concatenate((np.ones(bs), -np.ones(bs))))
            loss_gan = gan_train.train_on_batch(lr, [hr, np.ones(bs)])
        ref = orc.train_step(torch.tensor(lr, dtype=torch.float64), torch.tensor(hr, dtype=torch.float64))
        got = (loss_disc,) + tuple(loss_gan)
        for name, a, b in zip(("disc", "gan", "content", "adv"), got, ref):
            err = abs(a - b) / (abs(b) + 1e-6)
            report("train_step %s/%s/%s k%d it=%d loss_%s got=%.6g ref=%.6g rel=%.1e" % (wiring, losses, disc, k, it, name, a, b, err))
            assert err < 2e-3, (name, a, b)
    assert opt.iterations == orc.opt.iterations == 4
    # weights after two Adam steps each.  Adam normalises the step, so a parameter whose gradient is
    # pure rounding noise (conv biases in front of a BatchNorm: exactly zero in exact arithmetic) moves
    # by ~lr in an arbitrary direction in ANY fp32 implementation; those are compared only loosely.
    worst = 0.0
    for model, ow, tag in ((G, orc.g_w, "G"), (D, orc.d_w, "D")):
        got = model.get_weights_dict()
        for name, refv in ow.items():
            a, b = got[name].astype(np.float64), refv.detach().numpy()
            scale = np.max(np.abs(b)) + 1e-12
            err = float(np.max(np.abs(a - b)) / scale)
            noise_bias = name.endswith("/bias") and ("conv_pre" in name or "conv_post" in name or "prefinal" in name
                                                      or "/Conv2d/" in name or "Dense_1" in name or "Dense_2" in name)
            if noise_bias and disc != "patch" or (noise_bias and tag == "G"):
                assert np.max(np.abs(a - b)) < 5e-3, (name, err)
                continue
            worst = max(worst, err)
            assert err < 2e-3, (tag, name, err)
    report("train_step %s/%s/%s k%d worst weight err=%.2e" % (wiring, losses, disc, k, worst))


def test_generator_gradients_direct(rt):
    """dL/dtheta of the generator under a pixel MSE loss, compared tensor by tensor with autograd."""
    from oracle import models as M
    from upscaler import _engine as E, _lib as L
    gw, dw, gf, df = _oracle_pair(3, 2, "simple")
    G, D = _product_pair(3, 2, "simple", gw, dw)
    x, t = _frames(1, 2, 64, 64), _frames(2, 2, 128, 128)
    leaf = M.to_torch(gw, torch.float64, requires_grad=True)
    y, _ = gf(leaf, torch.tensor(x, dtype=torch.float64), True)
    loss = ((y - torch.tensor(t, dtype=torch.float64)) ** 2).mean()
    names = [k for k, v in leaf.items() if v.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(loss, [leaf[k] for k in names])))
    yd, tape = G.forward(E.to_device_nchw(rt, x), True)
    val, dy = rt.empty(1), rt.empty(*yd.shape)
    ws, wsn = rt.workspace(4096)
    L.check(rt.lib.vcg_pixel_loss(yd.data_ptr(), E.to_device_nchw(rt, t).data_ptr(), yd.numel(), L.LOSS_MSE, 1.0, val.data_ptr(),
                                  dy.data_ptr(), ws, wsn, rt.stream), "pixel_loss")
    G.backward(tape, dy, 0)
    gmax = max(float(g.abs().max()) for g in grads.values())
    worst = 0.0
    for k in names:
        a, b = G.ps.grad(k).cpu().double(), grads[k]
        # relative to the tensor's own scale, with a floor for gradients that are numerically zero
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6 * gmax))
        worst = max(worst, err)
        report("ggrad %-40s |g|=%.2e err=%.2e" % (k, float(b.abs().max()), err))
        assert err < 5e-3, (k, err)
    assert abs(val.item() - loss.item()) / loss.item() < 1e-4
