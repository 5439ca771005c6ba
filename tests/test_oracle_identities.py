"""Oracle self-consistency: identities that need no reference output."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import keras_ops as K
from oracle import models as M
from oracle import train as T

torch.manual_seed(0)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_conv_transpose_is_adjoint_of_same_strided_conv(k):
    # <conv_s2_same(x), y> == <x, convT_same(y)> with the Keras kernel mapping (Appendix A)
    cin, cout = 3, 4
    x = torch.randn(2, cin, 12, 16, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, dtype=torch.float64)           # Conv2D HWIO
    y = torch.randn(2, cout, 6, 8, dtype=torch.float64)
    lhs = (K.conv2d(x, w, None, 2, "same") * y).sum()
    # Conv2DTranspose kernel (kh,kw,out,in) with out=cin, in=cout is the same array
    rhs = (x * K.conv2d_transpose_same(y, w, None, 2)).sum()
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)


def test_same_conv_stride1_matches_symmetric_padding():
    x = torch.randn(1, 5, 9, 11, dtype=torch.float64)
    for k in (3, 5, 9):
        w = torch.randn(k, k, 5, 7, dtype=torch.float64)
        b = torch.randn(7, dtype=torch.float64)
        ref = F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=k // 2)
        assert torch.allclose(K.conv2d(x, w, b, 1, "same"), ref, atol=1e-12)


def test_batchnorm_properties():
    x = torch.randn(4, 6, 5, 7, dtype=torch.float64) * 3 + 2
    g, b = torch.rand(6, dtype=torch.float64) + 0.5, torch.randn(6, dtype=torch.float64)
    mm, mv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
    y, nmm, nmv = K.batchnorm(x, None, None, mm, mv, True)
    assert y.mean(dim=(0, 2, 3)).abs().max() < 1e-12
    var = x.var(dim=(0, 2, 3), unbiased=False)
    assert torch.allclose(y.var(dim=(0, 2, 3), unbiased=False), var / (var + 1e-3), atol=1e-12)
    m = 4 * 5 * 7
    assert torch.allclose(nmm, 0.01 * x.mean(dim=(0, 2, 3)))
    assert torch.allclose(nmv, 0.99 + 0.01 * var * m / (m - 1))       # fused path: Bessel-corrected
    # inference BN == folded affine
    yi, _, _ = K.batchnorm(x, g, b, nmm, nmv, False)
    sc = g / torch.sqrt(nmv + 1e-3)
    assert torch.allclose(yi, x * sc.view(1, -1, 1, 1) + (b - nmm * sc).view(1, -1, 1, 1), atol=1e-12)
    # 2-D (Dense) BN reports the biased variance
    x2 = torch.randn(8, 5, dtype=torch.float64)
    _, _, v2 = K.batchnorm(x2, None, None, torch.zeros(5, dtype=torch.float64), torch.ones(5, dtype=torch.float64), True)
    assert torch.allclose(v2, 0.99 + 0.01 * x2.var(dim=0, unbiased=False))


def test_prelu_zero_alpha_is_relu_and_flatten_order():
    x = torch.randn(2, 3, 4, 5)
    assert torch.equal(K.prelu(x, torch.zeros(3)), torch.relu(x))
    f = K.flatten_nhwc(x)
    assert f[1, (2 * 5 + 3) * 3 + 1] == x[1, 1, 2, 3]               # (h,w,c)-major


def test_adam_keras_form():
    p, g = torch.tensor([1.0, -2.0], dtype=torch.float64), torch.tensor([0.5, 1e-9], dtype=torch.float64)
    m, v = torch.zeros(2, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)
    p1, m1, v1 = K.adam_keras_step(p, g, m, v, 1)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert torch.allclose(m1, 0.1 * g) and torch.allclose(v1, 0.001 * g * g)
    assert torch.allclose(p1, p - lr_t * m1 / (torch.sqrt(v1) + 1e-7))
    # epsilon outside the bias correction: a 1e-9 gradient moves the weight by much less than lr
    assert abs(p1[1] - p[1]) < 0.3e-3 and abs(p1[0] - p[0]) > 0.99e-3


def test_loss_activations():
    x = torch.tensor([-3.0, 0.0, 2.5], dtype=torch.float64)
    assert torch.allclose(K.head_activation(x, "log-sigm"), torch.log(torch.sigmoid(x)))
    assert torch.allclose(K.head_activation(x, "bi-log"), x / (1 + x.abs()) * torch.log(x.abs() + 2))
    assert torch.equal(K.head_activation(x, "none"), x) and torch.equal(K.head_activation(x, "log"), x)  # Appendix D


def _tiny():
    gw = M.to_torch(M.init_upscaler_orig((32, 32, 3), 3, 64, 2, 1), torch.float64)
    dw = M.to_torch(M.init_discriminator_512((32, 32, 3), "thin"), torch.float64)
    gf = lambda w, x, t: M.upscaler_orig_forward(w, x, t, 1, 2)
    df = lambda w, x, t: M.discriminator_512_forward(w, x, t)
    return gf, gw, df, dw


def test_train_step_wiring_facts():
    gf, gw, df, dw = _tiny()
    o = T.GanOracle(gf, gw, df, dw)
    rng = np.random.RandomState(0)
    lr = torch.tensor(rng.randint(0, 256, (2, 16, 16, 3)) / 127.5 - 1)
    hr = torch.tensor(rng.randint(0, 256, (2, 32, 32, 3)) / 127.5 - 1)
    g_before = {k: v.clone() for k, v in o.g_w.items()}
    d_before = {k: v.clone() for k, v in o.d_w.items()}
    fake = o.predict(lr)
    o.disc_train_on_batch(hr, fake)
    assert o.opt.iterations == 1
    assert all(torch.equal(o.g_w[k], g_before[k]) for k in g_before)              # disc_train updates D only
    assert any(not torch.equal(o.d_w[k], d_before[k]) for k in d_before)
    d_mid = {k: v.clone() for k, v in o.d_w.items()}
    out = o.gan_train_on_batch(lr, hr)
    assert o.opt.iterations == 2                                                   # shared Adam counter
    assert all(torch.equal(o.d_w[k], d_mid[k]) for k in d_mid)                     # frozen D incl. moving stats
    assert abs(out[0] - (out[1] + 1e-5 * out[2])) < 1e-12
    # predict (moving stats) differs from the training-mode forward (batch stats)
    assert (o.last_fake_train - fake).abs().max() > 1e-4


def test_v1_wiring_loss_is_signed_mean():
    gf, gw, df, dw = _tiny()
    o = T.GanOracle(gf, gw, df, dw, wiring="v1")
    hr, fake = torch.randn(2, 32, 32, 3, dtype=torch.float64), torch.randn(2, 32, 32, 3, dtype=torch.float64)
    with torch.no_grad():
        d, _ = df(o.d_w, torch.cat([hr, fake]), True)
    expect = float((d[:2].sum() - d[2:].sum()) / 4)
    assert abs(o.disc_train_on_batch(hr, fake) - expect) < 1e-12
