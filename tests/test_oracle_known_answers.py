"""Pins the oracle against every known answer the reference's own notebooks hold for this path
 -- shapes, parameter counts, value map.  The reference has no numeric golden
vectors ("parity unpinned"), so these are the only reference-derived pins."""
import math

import numpy as np
import pytest
import torch

from oracle import data as OD
from oracle import keras_ops as K
from oracle import models as M


def _n(w):
    return int(sum(int(np.prod(v.shape)) for v in w.values()))


def test_cnn_test_cell12_shapes_and_params():
    # upscaling/cnn_test.ipynb cell 12: Input (135,240,3) -> Conv2D(1,k3,s2,same) -> (68,120,1), 28 params
    w, rng = {}, np.random.RandomState(0)
    M._conv_w(w, rng, "c", 3, 3, 3, 1)
    assert _n(w) == 28
    x = torch.zeros(1, 3, 135, 240)
    y = K.conv2d(x, torch.tensor(w["c/kernel"]), torch.tensor(w["c/bias"]), 2, "same")
    assert tuple(y.shape[2:]) == (68, 120)
    # 4x Conv2DTranspose(1,k3,s2,same): (136,240) -> (272,480) -> (544,960) -> (1088,1920), 10 params each
    total = 28
    for expect in ((136, 240), (272, 480), (544, 960), (1088, 1920)):
        wt = {}
        M._convt_w(wt, rng, "t", 3, 3, 1, 1)
        assert _n(wt) == 10
        total += 10
        y = K.conv2d_transpose_same(y, torch.tensor(wt["t/kernel"]), torch.tensor(wt["t/bias"]), 2)
        assert tuple(y.shape[2:]) == expect
    assert total == 68
    # cell 8: Cropping2D((4,0)) -> (1080,1920)
    assert ((1088 - 1080) // 2, (1920 - 1920) // 2) == (4, 0)


def test_cnn_test_cell18_param_counts():
    rng = np.random.RandomState(0)

    def conv(k, cin, cout):
        w = {}
        M._conv_w(w, rng, "c", k, k, cin, cout)
        return _n(w)

    def convt(k, cin, cout):
        w = {}
        M._convt_w(w, rng, "c", k, k, cin, cout)
        return _n(w)

    assert conv(9, 3, 128) == 31232
    w = {}
    M._prelu_w(w, "p", 128)
    assert _n(w) == 128                      # PReLU(shared_axes=[1,2]) on 128 channels
    assert conv(3, 128, 128) == 147584
    assert conv(3, 128, 256) == 295168
    assert conv(3, 256, 256) == 590080
    assert conv(3, 256, 512) == 1180160
    assert conv(3, 512, 512) == 2359808
    assert convt(3, 512, 512) == 2359808
    assert conv(3, 768, 256) == 1769728
    assert conv(9, 128, 3) == 31107
    # Conv2DTranspose(512,k3,s2): (34,60) -> (68,120) -> (136,240)
    y = torch.zeros(1, 2, 34, 60)
    wt = torch.zeros(3, 3, 2, 2)
    y = K.conv2d_transpose_same(y, wt, None, 2)
    assert tuple(y.shape[2:]) == (68, 120)
    assert tuple(K.conv2d_transpose_same(y, wt, None, 2).shape[2:]) == (136, 240)


def test_same_padding_rule():
    assert K.same_pads(135, 3, 2) == (68, 1, 1)
    assert K.same_pads(240, 3, 2) == (120, 0, 1)      # even input, k3 s2: pad (0,1)
    assert K.same_pads(512, 5, 2) == (256, 1, 2)
    assert K.same_pads(256, 9, 1) == (256, 4, 4)
    assert K.same_pads(256, 5, 1) == (256, 2, 2)


def test_model_param_counts_appendix_b():
    assert M.count_params(M.init_upscaler_orig((128, 128, 3), 3, 64, 2, 6)) == 709379
    assert M.count_params(M.init_upscaler_orig((128, 128, 3), 5, 64, 2, 6)) == 1823491
    assert M.count_params(M.init_upscaler_orig((512, 512, 3), 3, 64, 2, 9)) == 932675
    assert M.count_params(M.init_upscaler_orig((512, 512, 3), 5, 64, 2, 9)) == 2440003
    assert M.count_params(M.init_upscaler_orig((512, 512, 3), 5, 64, 4, 16)) == 5517187
    assert M.count_params(M.init_discriminator_512((128, 128, 3), "simple")) == 13926465
    assert M.count_params(M.init_discriminator_512((512, 512, 3), "simple")) == 15499329
    assert M.count_params(M.init_discriminator_512((1080, 1920, 3), "simple")) == 34373697
    assert M.count_params(M.init_discriminator_512((512, 512, 3), "thin")) == 1675457
    assert M.count_params(M.init_discriminator_patchgan_70((512, 512, 3))) == 2764737


def test_gan_test_output_shapes():
    # upscaling/gan_test.ipynb cells 17-18: discriminator output (N,1) float32; generator x2 upscale
    dw = M.to_torch(M.init_discriminator_512((128, 128, 3), "thin"))
    x = torch.zeros(3, 128, 128, 3)
    y, _ = M.discriminator_512_forward(dw, x, False)
    assert tuple(y.shape) == (3, 1) and y.dtype == torch.float32
    gw = M.to_torch(M.init_upscaler_orig((64, 64, 3), 3, 64, 2, 1))
    g, _ = M.upscaler_orig_forward(gw, torch.zeros(2, 32, 32, 3), False, 1, 2)
    assert tuple(g.shape) == (2, 64, 64, 3)
    p, _ = M.discriminator_patchgan_70_forward(M.to_torch(M.init_discriminator_patchgan_70((512, 512, 3))), torch.zeros(1, 512, 512, 3), True)
    assert tuple(p.shape) == (1, 62, 62, 1)


def test_value_map_data_py():
    # upscaling/upscaler/data.py:253-270 and minitrain_test.ipynb cells 7-8 (NHWC batches)
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    a = OD.convert_uint8_to_array(u8)
    assert a.dtype == np.float64 and a.shape == (1, 16, 16, 3)
    assert a.min() == -1.0 and a.max() == 1.0
    assert np.array_equal(a[0, :, :, 0].reshape(-1), np.arange(256) / 127.5 - 1)
    assert np.array_equal(OD.convert_array_to_uint8(a), u8)


def test_vgg19_notop_param_count_and_shapes():
    """keras.applications.VGG19(include_top=False): 20,024,384 parameters; block5_conv4 is 1/16 resolution x 512
    (the feature extractor of VGG_LOSS, upscaling/upscaler/model.py:108-112)."""
    import torch
    from oracle import models as M
    w = M.init_vgg19_features()
    assert sum(int(np.prod(v.shape)) for v in w.values()) == 20024384
    assert len(w) == 32 and w["block1_conv1/kernel"].shape == (3, 3, 3, 64) and w["block5_conv4/kernel"].shape == (3, 3, 512, 512)
    f = M.vgg19_block5_conv4(M.to_torch(w), torch.zeros(1, 32, 48, 3))
    assert tuple(f.shape) == (1, 2, 3, 512) and float(f.min()) >= 0.0


def test_sparse_512_and_attention_generator_sizes():
    """make_discriminator_sparse_512 at 512x512 has 5 987 777 parameters, counted layer by layer from model.py:964-1012 (SURVEY.md
    Appendix B says 5 987 137: it leaves out the 4 x (128 + 32) parameters of the Dense head's two BatchNormalizations, :999,:1003; spatial 512 -> 508 -> 168 -> 55 ->
    17 -> 5 -> 1); the attention generator's parameter count follows from model.py:30-48,78-98,299-328."""
    from oracle import models as M
    w = M.init_discriminator_sparse_512((512, 512, 3))
    assert M.count_params(w) == 5987777
    assert w["discriminator/final/Dense_1/kernel"].shape == (256, 128)
    y, _ = M.discriminator_sparse_512_forward(M.to_torch(w), torch.zeros(1, 512, 512, 3), False)
    assert tuple(y.shape) == (1, 1)
    k, f, res, c = 3, 64, 2, 3
    gw = M.init_upscaler_attention((64, 64, 3), k, f, 2, res)
    conv = lambda kk, ci, co: kk * kk * ci * co + co
    bn = 4 * f
    expect = conv(9, c, f) + f + res * (conv(k, c, f) + 2 * conv(k, f, f) + 2 * bn + f) + conv(k, f, f) + bn \
        + conv(k, 2 * c, f) + conv(k, f, 128) + conv(3, c, 128) + conv(9, 128, 3)
    assert M.count_params(gw) == expect


# ---- the other generators behind train_gan3.py -gm (oracle/generators.py) ---------------------------------------------------------
def test_skip_con_parameter_count_and_the_reference_name_clash():
    """make_upscaler_skip_con (model.py:332-363) at its defaults, counted by hand: 9x9 3->64 conv 15 616 + PReLU 64; 16 residual blocks of
    2 x (5*5*64*64 + 64) + 2 x 256 (BN: gamma, beta, moving mean / variance) + 64 = 205 504; 3x3 conv 36 928 + BN 256; Conv2DTranspose 3x3
    64->224 129 248 and 224->224 451 808; 9x9 conv (3 + 224)->3 55 164.  As written the reference cannot build it: sixteen layers named
    '/conv_pre' (keras.engine.network refuses duplicate names); the restatement raises the same way."""
    from oracle import generators as G, models as M
    with pytest.raises(ValueError, match="unique"):
        G.init_weights(G.upscaler_skip_con, (8, 8, 3), 1)
    w = G.init_weights(G.upscaler_skip_con, (8, 8, 3), 1, unique_names=True)
    assert M.count_params(w) == 15616 + 64 + 16 * 205504 + 36928 + 256 + 129248 + 451808 + 55164 == 3977148
    assert w["conv2d_3/kernel"].shape == (9, 9, 227, 3)                 # Keras' automatic names, counted per class from 1
    assert "batch_normalization_1/gamma" in w and "p_re_lu_1/alpha" in w


def _unetish_count(k, factor, step, down, c0, concat, halve):
    """independent arithmetic over model.py:577-609: (cin, cout) of every Conv2D / Conv2DTranspose + BN (4c) + PReLU (c) of the U"""
    total = 81 * 3 * c0 + c0 + c0
    blk = lambda ci, co: k * k * ci * co + co + 4 * co + co
    c, skips = c0, []
    cur = c0
    for _ in range(down):
        for _ in range(step):
            total += blk(cur, c); cur = c
        skips.append(cur)
        total += blk(cur, c); cur = c
        c *= 2
    for _ in range(step):
        total += blk(cur, c); cur = c
    if halve:
        c //= 2
    ups = int(math.log2(factor)) + down
    for s_ in range(ups):
        total += blk(cur, c); cur = c
        if s_ < len(skips):
            cur = cur + skips[len(skips) - s_ - 1] if concat else cur
            c //= 2
        for _ in range(step):
            total += blk(cur, c); cur = c
    return total, cur


def test_unetish_parameter_counts_and_shapes():
    from oracle import generators as G, models as M
    # reference defaults (kernel 5, x4, step size 4, five / five / three down-samplings)
    w = G.init_weights(G.upscaler_unetish, (32, 32, 3), 1)
    body, cl = _unetish_count(5, 4, 4, 5, 32, True, False)
    assert M.count_params(w) == body + 81 * cl * 3 + 3
    w = G.init_weights(G.upscaler_unetish_add, (32, 32, 3), 1)
    body, cl = _unetish_count(5, 4, 4, 5, 48, False, True)
    assert M.count_params(w) == body + (81 * cl * 3 + 3) + (81 * 3 * 3 + 3)
    w = G.init_weights(G.upscaler_unetish_complex, (32, 32, 3), 1)
    body, cl = _unetish_count(5, 4, 4, 3, 32, True, False)
    head = (81 * 3 * 3 + 3) + (81 * 6 * 3 + 3) * 3 + (81 * cl * 3 + 3) + 2 * (81 * 3 * 3 + 3) + 3 * (81 * 6 * 3 + 3)
    assert M.count_params(w) == body + head
    # an odd frame: 11 -> 6 -> 12 is cropped back to 11 at the join, the output is exactly factor x input
    x = torch.rand(1, 22, 30, 3, dtype=torch.float64)
    kw = dict(kernel_size=3, upscale_factor=2, step_size=1, downscale_times=2, initial_step_filter_count=8)
    for fn in (G.upscaler_unetish, G.upscaler_unetish_add, G.upscaler_unetish_complex):
        wf = G.init_weights(fn, (22, 30, 3), 2, **kw)
        with torch.no_grad():
            y = fn(G.Net(M.to_torch(wf, torch.float64), False), x, **kw)
        assert y.shape == (1, 44, 60, 3) and float(y.abs().max()) <= 1.0


def test_dropout_restatement_is_identity_outside_training_and_scales_inside():
    from oracle import generators as G
    x = torch.arange(12, dtype=torch.float64).view(1, 3, 2, 2)
    mask = (torch.arange(12).view(1, 3, 2, 2) % 3 != 0)
    assert torch.equal(G.Net({}, False).dropout(x, 0.25, "d"), x)                       # learning phase 0
    assert torch.equal(G.Net({}, True).dropout(x, 0.0, "d"), x)                         # keras.layers.Dropout.call: 0 < rate < 1
    y = G.Net({}, True, masks={"d": mask}).dropout(x, 0.25, "d")
    assert torch.equal(y, torch.where(mask, x / 0.75, torch.zeros_like(x)))


def test_incep_resnet_parameter_count():
    """make_upscaler_incep_resnet (model.py:443-497) at its defaults, counted by hand.  Per mini block: BN 4c + PReLU c on the INPUT
    channels, then the convolution.  3-path (k=3, f=64): a 64->32 1x1; b 64->32 1x1, 32->32 3x3; c 64->32 1x1, 32->48 3x3, 48->64 3x3;
    concat 128 -> 64 1x1.  2-path (k, f=64): a 64->32 1x1; b 64->19 1x1, 19->25 1xk, 25->32 kx1; concat 64 -> 64 1x1."""
    from oracle import generators as G, models as M
    mini = lambda ci, co, t: 5 * ci + t * ci * co + co
    p3 = lambda k: (mini(64, 32, 1) + mini(64, 32, 1) + mini(32, 32, k * k) + mini(64, 32, 1) + mini(32, 48, k * k) + mini(48, 64, k * k)
                    + 128 * 64 + 64)
    p2 = lambda k: mini(64, 32, 1) + mini(64, 19, 1) + mini(19, 25, k) + mini(25, 32, k) + 64 * 64 + 64
    expected = (81 * 3 * 64 + 64) + 5 * p3(3) + 10 * p2(7) + 5 * p2(3) + (9 * 64 * 64 + 64) + 256 \
        + (9 * 64 * 256 + 256) + (9 * 256 * 256 + 256) + (81 * 256 * 3 + 3)
    w = G.init_weights(G.upscaler_incep_resnet, (8, 8, 3), 1)
    assert M.count_params(w) == expected
    assert w["inc_res_block/B/2p/0/b/2/1x7/kernel"].shape == (1, 7, 19, 25) and w["inc_res_block/B/2p/0/b/3/7x1/kernel"].shape == (7, 1, 25, 32)
    assert "inc_res_block/c/2p/4/final/1x1/kernel" in w                       # the third group is tagged 'c' in lower case (model.py:479,481)
