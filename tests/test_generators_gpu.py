"""The other generator topologies train_gan3.py offers behind -gm (upscaling/train_gan3.py:55,234-252; upscaler/model.py:332-363,
505-827) on the device engine against their fp64 oracle restatements (oracle/generators.py): parameter count and names, inference,
learning-phase forward (Dropout: same keep-masks on both sides), every parameter gradient, moving statistics.  Frame sizes are odd
where the topology crops (U-Net-ish joins: 11 -> 6 -> 12 -> crop 11)."""
import numpy as np
import pytest
import torch

from conftest import rel_err, report

pytestmark = pytest.mark.gpu


def _randomise(gw, seed):
    rng = np.random.RandomState(seed)
    for n_, v in gw.items():                              # non-trivial BN / PReLU parameters
        if n_.endswith(("/bias", "/beta", "/moving_mean")):
            gw[n_] = rng.uniform(-0.1, 0.1, v.shape).astype(np.float32)
        elif n_.endswith(("/gamma", "/moving_variance")):
            gw[n_] = rng.uniform(0.8, 1.2, v.shape).astype(np.float32)
        elif n_.endswith("/alpha"):
            gw[n_] = rng.uniform(0.0, 0.3, v.shape).astype(np.float32)
    return gw


def _prelu_masks(product, tape):
    """{PReLU layer name: mask (pre-activation >= 0)} as the device evaluated them in the last training forward"""
    from upscaler import _lib as L
    out = {}
    for name, (layer, ctx) in product.norm_contexts(tape).items():
        if layer.act != L.ACT_PRELU:
            continue
        x, saved, mode, (n, c, hw) = ctx
        z = x
        if saved is not None:
            mean, invstd = saved
            rows = mean.numel() // c
            z = (x - mean.view(rows, c, 1, 1)) * invstd.view(rows, c, 1, 1)
            if layer.norm == "batch":
                z = z * layer.ps[layer.name + "/gamma"].view(1, c, 1, 1) + layer.ps[layer.name + "/beta"].view(1, c, 1, 1)
        out[layer.prelu_name] = (z >= 0).cpu()
    return out


def _check(rt, tag, product, oracle_fn, okw, in_hw, out_hw, seed, device_prelu_masks=False):
    from oracle import generators as OG, models as M
    from upscaler import _engine as E, model as PM
    gw = _randomise(OG.init_weights(oracle_fn, in_hw + (3,), seed, **okw), seed + 1)
    assert product.count_params() == M.count_params(gw)
    assert set(product.get_weights_dict()) == set(gw)
    assert product.input_shape == (None,) + in_hw + (3,) and product.output_shape == (None,) + out_hw + (3,)
    product.set_weights_dict(gw)
    rng = np.random.RandomState(seed + 2)
    x = (rng.randint(0, 256, (2,) + in_hw + (3,)) / 127.5 - 1).astype(np.float32)
    t = (rng.randint(0, 256, (2,) + out_hw + (3,)) / 127.5 - 1).astype(np.float32)
    with torch.no_grad():
        y0 = oracle_fn(OG.Net(M.to_torch(gw, torch.float64), False), torch.tensor(x, dtype=torch.float64), **okw)
    e0 = rel_err(torch.tensor(product.predict(x)), y0)

    yd, tape = product.forward(E.to_device_nchw(rt, x), True)
    masks = {k: v.cpu() for k, v in product.dropout_masks(tape).items()}
    for k, m in masks.items():
        keep = float(m.float().mean())
        assert 0.8 < keep < 0.97 or m.numel() < 2000, (k, keep)          # rate 0.1
    ndrop = len(masks)
    if device_prelu_masks:
        # the gradient check runs on the branch of the (piecewise-linear) network the device took: a pre-activation within fp32 rounding of 0
        # gets either mask in any fp32 evaluation, and ONE such element moves a weight gradient on a 6x10 map by percents (iii)
        masks.update(_prelu_masks(product, tape))

    def run(dtype):
        leaf = M.to_torch(gw, dtype, requires_grad=True)
        net = OG.Net(leaf, True, masks=masks)
        y = oracle_fn(net, torch.tensor(x, dtype=dtype), **okw)
        names = [n_ for n_, v in leaf.items() if v.requires_grad]
        loss = ((y - torch.tensor(t, dtype=dtype)) ** 2).mean()
        return y, net.upd, names, dict(zip(names, torch.autograd.grad(loss, [leaf[n_] for n_ in names])))

    y, upd, names, grads = run(torch.float64)
    _, _, _, g32 = run(torch.float32)
    e1 = rel_err(E.to_nhwc(rt, yd), y)
    val, dy = PM._pixel_loss(rt, yd, E.to_device_nchw(rt, t), "mse", 1.0)
    product.backward(tape, dy, 0)
    gmax = max(float(g.abs().max()) for g in grads.values())
    worst, worst_name = 0.0, None
    for n_ in names:
        a, b = product.ps.grad(n_).cpu().double().reshape(grads[n_].shape), grads[n_]
        err = float((a - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
        e32 = float((g32[n_].double() - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
        if err > worst:
            worst, worst_name = err, n_
        assert err < max(1e-3, 4 * e32), (n_, err, e32)
    sw = product.get_weights_dict()
    for n_, v in upd.items():
        assert np.max(np.abs(sw[n_] - v.detach().numpy())) < 1e-4 * (np.max(np.abs(v.detach().numpy())) + 1e-3), n_
    report("%s: %d params, %d dropout layers; predict err=%.2e train fwd err=%.2e worst gradient err=%.2e (%s)"
           % (tag, product.count_params(), ndrop, e0, e1, worst, worst_name))
    assert e0 < 1e-3 and e1 < 1e-3


U = dict(kernel_size=3, upscale_factor=2, step_size=1, downscale_times=2, initial_step_filter_count=32, dropout_rate=0.1)


def test_unetish_matches_oracle(rt):
    from oracle import generators as OG
    from upscaler import model as PM
    _check(rt, "make_upscaler_unetish", PM.make_upscaler_unetish((44, 60, 3), **U), OG.upscaler_unetish, U, (22, 30), (44, 60), 3)


def test_unetish_add_matches_oracle(rt):
    from oracle import generators as OG
    from upscaler import model as PM
    _check(rt, "make_upscaler_unetish_add", PM.make_upscaler_unetish_add((44, 60, 3), **U), OG.upscaler_unetish_add, U, (22, 30), (44, 60), 5)


def test_unetish_complex_matches_oracle(rt):
    from oracle import generators as OG
    from upscaler import model as PM
    _check(rt, "make_upscaler_unetish_complex", PM.make_upscaler_unetish_complex((44, 60, 3), **U), OG.upscaler_unetish_complex, U, (22, 30), (44, 60), 7)


def test_unetish_x4_default_depth_shapes(rt):
    """the reference's defaults (kernel 5, x4, five down-samplings, step size 4) at a small frame: builds, crops to the requested output"""
    from upscaler import model as PM
    G = PM.make_upscaler_unetish((72, 104, 3), step_size=1)
    assert G.input_shape == (None, 18, 26, 3) and G.output_shape == (None, 72, 104, 3)
    x = (np.random.RandomState(0).randint(0, 256, (1, 18, 26, 3)) / 127.5 - 1).astype(np.float32)
    y = G.predict(x)
    assert y.shape == (1, 72, 104, 3) and np.isfinite(y).all() and np.abs(y).max() <= 1.0


def test_skip_con_matches_oracle_and_mirrors_the_reference_error(rt):
    from oracle import generators as OG
    from upscaler import model as PM
    with pytest.raises(ValueError, match="unique"):                     # what keras.engine.network does with sixteen '/conv_pre'
        PM.make_upscaler_skip_con((32, 48, 3), kernel_size=3, upscale_factor=2)
    kw = dict(kernel_size=3, upscale_factor=2, unique_names=True)
    _check(rt, "make_upscaler_skip_con", PM.make_upscaler_skip_con((32, 48, 3), **kw), OG.upscaler_skip_con, kw, (16, 24), (32, 48), 9)


def test_incep_resnet_matches_oracle(rt):
    """make_upscaler_incep_resnet (model.py:443-497) with one block of each kind the reference's defaults use: 3-path k3 (1x1, 3x3),
    2-path k7 (1x1, 1x7, 7x1 on 19- / 25- / 32-channel paths), 2-path k3 (1x3, 3x1)"""
    from oracle import generators as OG
    from upscaler import model as PM
    kw = dict(filters=64, upscale_factor=2, a_block_num=1, b_block_num=1, c_block_num=1)
    _check(rt, "make_upscaler_incep_resnet", PM.make_upscaler_incep_resnet((40, 72, 3), **kw), OG.upscaler_incep_resnet, kw, (20, 36), (40, 72), 11)


def test_incep_resnet_other_block_choices(rt):
    """3-path with kernel 5 and 2-path with kernel 5 (train_gan3.py --a_block_kernel ... :88-99): inference against the oracle"""
    from oracle import generators as OG, models as M
    from upscaler import model as PM
    kw = dict(filters=64, upscale_factor=2, a_block_type="2path", a_block_num=1, a_block_kernel=5, b_block_type="3path", b_block_num=1, b_block_kernel=5,
              c_block_num=0)
    G = PM.make_upscaler_incep_resnet((32, 48, 3), **kw)
    gw = _randomise(OG.init_weights(OG.upscaler_incep_resnet, (16, 24, 3), 4, **kw), 5)
    assert G.count_params() == M.count_params(gw)
    G.set_weights_dict(gw)
    x = (np.random.RandomState(6).randint(0, 256, (2, 16, 24, 3)) / 127.5 - 1).astype(np.float32)
    with torch.no_grad():
        y0 = OG.upscaler_incep_resnet(OG.Net(M.to_torch(gw, torch.float64), False), torch.tensor(x, dtype=torch.float64), **kw)
    e0 = rel_err(torch.tensor(G.predict(x)), y0)
    report("make_upscaler_incep_resnet (2-path k5 / 3-path k5): predict err=%.2e" % e0)
    assert e0 < 1e-3


@pytest.mark.parametrize("norm,f,n_down,res", [("instance", 1, 2, 3), ("batch", 2, 1, 2)])
def test_generator_cyclegan_matches_oracle(rt, norm, f, n_down, res):
    """make_generator_cyclegan -- BASELINE.json north_star's literal generator shape (down-sampling convolutions -> residual blocks ->
    transposed-convolution up-sampling) -- frame-to-frame with instance norm (256-channel residual blocks
    at h/4) and as a x2 up-scaler with BatchNormalization: names, parameter count, inference, training forward, every gradient."""
    from oracle import generators as OG
    from upscaler import model as PM
    h, w = 24, 40
    G = PM.make_generator_cyclegan((h * f, w * f, 3), filters=64, n_downsample=n_down, res_block_num=res, upscale_factor=f, norm=norm, seed=3)
    _check(rt, "cyclegan generator (%s norm, x%d, %d down, %d blocks)" % (norm, f, n_down, res), G, OG.generator_cyclegan,
           dict(filters=64, n_downsample=n_down, res_block_num=res, upscale_factor=f, norm=norm), (h, w), (h * f, w * f), 31,
           device_prelu_masks=True)


def test_dropout_masks_change_every_step_and_vanish_at_inference(rt):
    from upscaler import _engine as E, model as PM
    G = PM.make_upscaler_unetish((44, 60, 3), **U)
    x = E.to_device_nchw(rt, (np.random.RandomState(1).randint(0, 256, (2, 22, 30, 3)) / 127.5 - 1).astype(np.float32))
    _, t1 = G.forward(x, True)
    _, t2 = G.forward(x, True)
    m1, m2 = G.dropout_masks(t1), G.dropout_masks(t2)
    assert m1 and set(m1) == set(m2)
    assert all(not torch.equal(m1[k], m2[k]) for k in m1)
    _, t0 = G.forward(x, False)
    assert not G.dropout_masks(t0)
    a, b = G.predict(x.permute(0, 2, 3, 1).cpu().numpy()), G.predict(x.permute(0, 2, 3, 1).cpu().numpy())
    assert np.array_equal(a, b)


def test_unetish_gan_step_graph_replay_matches_eager(rt):
    """the three-call train step with a Dropout-bearing generator, eager against hipGraph replay: the keep-masks come from a
    device-side step counter, so a recorded graph draws the same sequence of fresh masks as the eager loop -- bit for bit"""
    from upscaler import _engine as E, model as PM
    frames = [((np.random.RandomState(30 + i).randint(0, 256, (2, 32, 32, 3)) / 127.5 - 1).astype(np.float32),
               (np.random.RandomState(40 + i).randint(0, 256, (2, 64, 64, 3)) / 127.5 - 1).astype(np.float32)) for i in range(4)]

    def run(graph):
        G = PM.make_upscaler_unetish((64, 64, 3), **U)
        D = PM.make_discriminator_patchgan_70((64, 64, 3), seed=11)
        opt = PM.Adam()
        _, _, gan_train = PM.make_and_compile_gan2(G, D, (32, 32, 3), (64, 64, 3), "mse", 1.0, lambda: PM.WassersteinLosses(), 1e-2, optimizer=opt)
        tr = gan_train.trainer
        dev = [(E.to_device_nchw(rt, a), E.to_device_nchw(rt, b)) for a, b in frames]
        out = []
        if graph:
            tr.capture_train_step(*dev[0])
            for a, b in dev[1:]:
                out.append(tr.train_step_graph(a, b))
        else:
            tr._t_dev = torch.tensor([opt.iterations, 0], dtype=torch.int32, device=rt.device)
            tr.train_step(*dev[0])
            for a, b in dev[1:]:
                out.append(tr.train_step(a, b))
        return out, G.get_weights_dict(), int(G._drop_step.item())

    oe, ge, se = run(False)
    og, gg, sg = run(True)
    assert se == sg and se >= 4                                  # one mask draw per training forward of G, eager and replayed alike
    for a, b in zip(oe, og):
        assert a == b, (a, b)
    assert all(np.array_equal(ge[k], gg[k]) for k in ge)
    assert len({o[0] for o in oe}) == len(oe)                    # the steps differ (fresh frames, fresh masks)
