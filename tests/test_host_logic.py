"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/vcg.h
declares, the Python mirror keeps the reference's names and signatures, and the product path fails
loudly without a GPU (no CPU fallback, no oracle import)."""
import inspect
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-cycle_gan-upscaling_amd")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from upscaler import _lib
    return _lib


def _declared_in_header():
    src = open(os.path.join(ROOT, "include", "vcg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(vcg_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol(built):
    lib = built.load()
    declared = _declared_in_header()
    assert declared == set(built.SIGNATURES), declared ^ set(built.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.vcg_version()
    assert lib.vcg_error_string(-3).decode().startswith("kernel size")


def test_workspace_queries_and_argument_errors_without_gpu(built):
    """pure host logic of the ABI: size queries and invalid-argument codes need no device"""
    import ctypes
    lib = built.load()
    d = built.ConvDesc(8, 64, 256, 256, 64, 256, 256, 3, 3, 1, 1, 1)
    ws = lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
    assert 0 < ws < (1 << 30)
    assert lib.vcg_norm_stats_workspace_bytes(8, 64, 65536, 0) > 0
    assert lib.vcg_conv2d_fwd(None, None, None, None, None, None) == -1           # VCG_E_NULL
    bad = built.ConvDesc(8, 64, 256, 256, 64, 256, 256, 3, 3, 4, 1, 1)            # stride 4: not instantiated
    assert lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(bad)) == 0
    assert lib.vcg_conv2d_dgrad(ctypes.byref(bad), None, None, None, None, None, None) == -3       # VCG_E_UNSUPPORTED before any pointer is touched
    # sparse_512's layer (model.py:971): 5x5, stride 3, 'valid'
    sp = built.ConvDesc(8, 64, 508, 508, 128, 168, 168, 5, 5, 3, 0, 0)
    assert lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(sp)) > 0
    # generic bf16 convolution: PatchGAN block 4 (256 -> 512, 4x4, stride 1, pad 1)
    g = built.ConvDesc(8, 256, 64, 64, 512, 63, 63, 4, 4, 1, 1, 1)
    assert lib.vcg_conv2d_nhwc_bf16_wgrad_workspace_bytes(ctypes.byref(g)) > 0
    assert lib.vcg_conv_frag_bf16_bytes(16, 512, 256) == 16 * 512 * 256 * 2
    odd = built.ConvDesc(8, 48, 64, 64, 512, 63, 63, 4, 4, 1, 1, 1)               # 48 input channels: not a multiple of 64
    assert lib.vcg_conv2d_nhwc_bf16_wgrad_workspace_bytes(ctypes.byref(odd)) == 0
    assert lib.vcg_conv2d_nhwc_bf16_fwd(None, None, None, None, 0, 0.0, None, None) == -1
    assert lib.vcg_head_act_fwd(None, None, 4, 1, None) == -1
    assert lib.vcg_gan_loss(None, None, 1.0, 0, None, None, 0, 0.0, None, 0, 0.0, None) == -1
    with pytest.raises(ValueError):
        built.check(-2, "x")


def test_reference_signatures_are_mirrored():
    from upscaler import model as PM

    def params(f):
        return [(p.name, p.default) for p in inspect.signature(f).parameters.values()]

    # upscaling/upscaler/model.py:267
    assert params(PM.make_upscaler_orig)[:5] == [("output_image_shape", inspect._empty), ("kernel_size", 5), ("filters", 64),
                                                 ("upscale_factor", 4), ("res_block_num", 16)]
    # :836, :901
    assert params(PM.make_discriminator_simple_512)[:2] == [("input_shape", inspect._empty), ("activation", "none")]
    assert params(PM.make_discriminator_thin_512)[:2] == [("input_shape", inspect._empty), ("activation", "none")]
    # :964 and :299
    assert params(PM.make_discriminator_sparse_512)[:2] == [("input_shape", inspect._empty), ("activation", "none")]
    assert params(PM.make_upscaler_attention)[:5] == [("output_image_shape", inspect._empty), ("kernel_size", 5), ("filters", 64),
                                                      ("upscale_factor", 4), ("res_block_num", 16)]
    # the other -gm choices: :332, :570, :642, :743 and their blocks :505-566
    assert params(PM.make_upscaler_skip_con)[:4] == [("output_image_shape", inspect._empty), ("kernel_size", 5), ("filters", 64), ("upscale_factor", 4)]
    assert params(PM.make_upscaler_unetish)[:7] == [("output_image_shape", inspect._empty), ("kernel_size", 5), ("upscale_factor", 4), ("step_size", 4),
                                                    ("downscale_times", 5), ("initial_step_filter_count", 32), ("dropout_rate", 0.1)]
    assert params(PM.make_upscaler_unetish_add)[:7] == [("output_image_shape", inspect._empty), ("kernel_size", 5), ("upscale_factor", 4), ("step_size", 4),
                                                        ("downscale_times", 5), ("initial_step_filter_count", 48), ("dropout_rate", 0.1)]
    assert params(PM.make_upscaler_unetish_complex)[:7] == [("output_image_shape", inspect._empty), ("kernel_size", 5), ("upscale_factor", 4),
                                                            ("step_size", 4), ("downscale_times", 3), ("initial_step_filter_count", 32), ("dropout_rate", 0.1)]
    # :443-449
    assert params(PM.make_upscaler_incep_resnet)[:12] == [("output_image_shape", inspect._empty), ("filters", 64), ("upscale_factor", 4),
                                                          ("a_block_type", "3path"), ("a_block_num", 5), ("a_block_kernel", 3),
                                                          ("b_block_type", "2path"), ("b_block_num", 10), ("b_block_kernel", 7),
                                                          ("c_block_type", "2path"), ("c_block_num", 5), ("c_block_kernel", 3)]
    for f in (PM.same_size_unetish_block, PM.downsampling_unetish_block, PM.upsampling_unetish_block):
        assert params(f) == [("model", inspect._empty), ("kernel_size", inspect._empty), ("filters", inspect._empty), ("strides", inspect._empty),
                             ("name", inspect._empty), ("dropout_rate", 0.1)]
    # the block functions, :30 and :78
    assert [n for n, _ in params(PM.residual_block_attention)][:6] == ["model", "input_", "kernel_size", "filters", "strides", "batch_norm"]
    assert [n for n, _ in params(PM.upsampling_block_attention)][:5] == ["model", "input_", "scale", "kernel_size", "filters"]
    # :1017-1027, :1057-1067
    names = [n for n, _ in params(PM.make_and_compile_gan)]
    assert names[:9] == ["generator", "discriminator", "input_shape", "output_shape", "content_loss", "content_loss_weight",
                         "discriminator_loss", "discriminator_loss_weight", "optimizer"]
    names = [n for n, _ in params(PM.make_and_compile_gan2)]
    assert names[:9] == ["generator", "discriminator", "input_shape", "output_shape", "content_loss", "content_loss_weight",
                         "discriminator_losses", "discriminator_loss_weight", "optimizer"]
    assert [n for n, _ in params(PM.compile_training_model)] == ["upscaler", "loss", "optimizer"]
    assert [n for n, _ in params(PM.wasserstein_loss)] == ["y_true", "y_pred"]
    # :166-261
    assert params(PM.GanLosses.__init__)[1:] == [("loss_activation", "log-sigm"), ("real_output", None), ("fake_output", None)]
    for cls in (PM.WassersteinLosses, PM.RelativisticLosses):
        for prop in ("real_output", "fake_output", "discriminator_loss", "generator_loss"):
            assert isinstance(getattr(cls, prop), property)
    for meth in ("predict", "save"):
        assert hasattr(PM.Model, meth)
    for meth in ("predict", "train_on_batch"):
        assert hasattr(PM.TrainingModel, meth)
    from upscaler import data as PD
    for fn in ("convert_array_to_image", "convert_image_to_array", "convert_image_series_to_array"):
        assert hasattr(PD, fn)


def test_host_side_losses_and_optimizer():
    from upscaler import model as PM
    w = PM.WassersteinLosses()
    w.real_output, w.fake_output = np.array([[1.0], [3.0]]), np.array([[0.5], [0.5]])
    assert w.discriminator_loss(None, None) == 1.5 and w.generator_loss(None, None) == 0.5
    r = PM.RelativisticLosses(loss_activation="log-sigm", real_output=np.array([2.0]), fake_output=np.array([0.0]))
    assert abs(r.discriminator_loss(None, None) - math.log(1 / (1 + math.exp(-2.0)))) < 1e-12
    assert abs(r.generator_loss(None, None) - math.log(1 / (1 + math.exp(2.0)))) < 1e-12
    assert PM.RelativisticLosses(loss_activation="log").loss_activation(0.3) == 0.3        # unknown name -> identity (Appendix D)
    for name in ("sigmoid", "log-sigm", "tanh", "bi-log", "none"):
        for x in (-2.3, -0.1, 0.4, 3.0):
            v, g = PM._act_value_and_grad(name, x)
            h = 1e-6
            fd = (PM._act_value_and_grad(name, x + h)[0] - PM._act_value_and_grad(name, x - h)[0]) / (2 * h)
            assert abs(g - fd) < 1e-6, (name, x)
    assert PM.wasserstein_loss(np.array([1, -1]), np.array([[2.0], [4.0]])) == -1.0
    a = PM.Adam()
    f32 = lambda v: float(np.float32(v))        # lr_t is evaluated on the float32 values the kernels (and Keras' graph) use
    assert abs(a.lr_t() - f32(1e-3) * math.sqrt(1 - f32(0.999)) / (1 - f32(0.9))) < 1e-15
    assert abs(a.lr_t() - 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)) < 1e-7 * a.lr_t() * 1e3
    assert PM._content_kind("mean_squared_error") == "mse" and PM._content_kind(PM.PixelLoss("mae").loss) == "mae"
    with pytest.raises(NotImplementedError):
        PM._content_kind(lambda a, b: 0)


def test_padding_arithmetic_matches_oracle():
    from oracle import keras_ops as K
    from upscaler import _engine as E
    for size in (2, 5, 64, 135, 240, 512, 1080):
        for k in (3, 4, 5, 9):
            for s in (1, 2):
                assert E.same_pads(size, k, s) == K.same_pads(size, k, s)


def test_data_value_map_host():
    from upscaler import data as PD
    u8 = np.random.RandomState(0).randint(0, 256, (2, 5, 6, 3)).astype(np.uint8)
    a = PD.convert_image_series_to_array(list(u8))
    assert a.dtype == np.float64 and np.array_equal(a, u8 / 127.5 - 1)
    img = PD.convert_array_to_image(a[0])
    assert np.array_equal(np.array(img), u8[0])


def test_product_fails_loudly_without_gpu_and_never_imports_oracle():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from upscaler import model as PM\n"
            "try:\n"
            "    PM.make_upscaler_orig((128,128,3), 3, 64, 2, 1)\n"
            "    print('NOFAIL')\n"
            "except RuntimeError as e:\n"
            "    print('RAISED', 'oracle' in sys.modules)\n") % PKG
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout
    assert "RAISED False" in out, out


def test_product_sources_do_not_reference_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_c_abi_argument_validation_without_a_gpu():
    """Error convention of the C ABI (include/vcg.h: 0 ok, <0 VCG_E_*, >0 hipError_t): argument checks come before any device
    work, so they can be exercised on a box without a GPU."""
    import ctypes
    from upscaler import _lib as L
    lib = L.load()
    E_NULL, E_SHAPE, E_UNSUPPORTED, E_WORKSPACE = -1, -2, -3, -4
    d = L.ConvDesc(1, 64, 8, 8, 64, 8, 8, 3, 3, 1, 1, 1)
    one = ctypes.c_void_p(16)                              # any non-null address: never dereferenced by the checks below
    # required pointers
    assert lib.vcg_conv2d_fwd(ctypes.byref(d), None, one, one, None, None) == E_NULL
    assert lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), one, None, one, None, None) == E_NULL
    assert lib.vcg_norm_stats_bf16(None, 1, 64, 64, 0, one, one, one, 1 << 20, None) == E_NULL
    # inconsistent shapes
    bad = L.ConvDesc(0, 64, 8, 8, 64, 8, 8, 3, 3, 1, 1, 1)
    assert lib.vcg_conv2d_fwd(ctypes.byref(bad), one, one, one, None, None) == E_SHAPE
    assert lib.vcg_conv2d_bf16_fwd(ctypes.byref(bad), one, one, one, None, None) == E_SHAPE
    assert lib.vcg_maxpool2x2_fwd(one, one, 1, 3, 1, 8, None) == E_SHAPE
    assert lib.vcg_pack_conv_kernel_bf16(one, 0, 64, 64, 1, 0, one, None) == E_SHAPE
    # shapes that are not instantiated
    d7 = L.ConvDesc(1, 64, 8, 8, 64, 8, 8, 7, 7, 1, 3, 3)
    assert lib.vcg_conv2d_fwd(ctypes.byref(d7), one, one, one, None, None) == E_UNSUPPORTED
    d128 = L.ConvDesc(1, 128, 8, 8, 64, 8, 8, 3, 3, 1, 1, 1)
    assert lib.vcg_conv2d_bf16_fwd(ctypes.byref(d128), one, one, one, None, None) == E_UNSUPPORTED
    assert lib.vcg_conv2d_bf16_wgrad(ctypes.byref(d128), one, one, one, None, one, 1 << 30, None) == E_UNSUPPORTED
    assert lib.vcg_norm_stats_bf16(one, 1, 60, 64, 0, one, one, one, 1 << 20, None) == E_UNSUPPORTED       # c % 8 != 0
    ep = L.EpilogueBf16(None, None, L.ACT_TANH, 0.0, None, None)
    assert lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), one, one, one, ctypes.byref(ep), None) == E_UNSUPPORTED
    ep = L.EpilogueBf16(None, None, L.ACT_PRELU, 0.0, None, None)                                          # PReLU without its slopes
    assert lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), one, one, one, ctypes.byref(ep), None) == E_NULL
    # workspace too small
    need = lib.vcg_conv2d_bf16_wgrad_workspace_bytes(ctypes.byref(d))
    assert need > 0 and lib.vcg_conv2d_bf16_wgrad(ctypes.byref(d), one, one, one, None, one, need - 1, None) == E_WORKSPACE
    need = lib.vcg_norm_act_bwd_bf16_workspace_bytes(2, 64, 64, 0)
    assert lib.vcg_norm_act_bwd_bf16(one, one, 2, 64, 64, 0, one, one, None, None, 0, 0.0, None, 1, one, None, None, None, one, need - 1, None) == E_WORKSPACE
    # the entry points behind the other generators (resize / crop / concat / dropout) and the later bf16 kernels
    assert lib.vcg_resize2d(None, one, 6, 8, 8, 2, 1, None) == E_NULL
    assert lib.vcg_resize2d(one, one, 6, 8, 8, 0, 1, None) == E_SHAPE
    assert lib.vcg_crop2d(one, one, 6, 8, 8, 4, 0, 8, 8, None) == E_SHAPE                  # the window leaves the source
    assert lib.vcg_pad2d(one, one, 6, 8, 8, 1, 1, 8, 8, None) == E_SHAPE                   # the source does not fit the target
    assert lib.vcg_copy_channels(one, one, 2, 3, 0, 227, 225, 3, 42, None) == E_SHAPE      # channel block past the destination
    assert lib.vcg_dropout_fwd(one, one, one, 16, 1.0, 1, None, None) == E_SHAPE           # rate must be in [0, 1)
    assert lib.vcg_dropout_bwd(one, None, one, 16, 0.1, None) == E_NULL
    assert lib.vcg_counter_inc(None, None) == E_NULL
    d9 = L.ConvDesc(1, 256, 8, 9, 3, 8, 9, 9, 9, 1, 4, 4)                                   # odd width
    need = lib.vcg_conv9x9_to3_bf16_wgrad_workspace_bytes(ctypes.byref(d9))
    assert need > 0 and lib.vcg_conv9x9_to3_bf16_wgrad(ctypes.byref(d9), one, one, one, one, need, None) == E_UNSUPPORTED
    d9 = L.ConvDesc(1, 256, 8, 8, 3, 8, 8, 9, 9, 1, 4, 4)
    assert lib.vcg_conv9x9_to3_bf16_wgrad(ctypes.byref(d9), one, one, one, one, 16, None) == E_WORKSPACE
    dt = L.ConvDesc(1, 64, 8, 8, 256, 8, 8, 3, 3, 1, 0, 0)                                  # a transposed convolution with strides 1
    assert lib.vcg_conv_transpose2d_nhwc_bf16_fwd(ctypes.byref(dt), one, one, None, 0, 0.0, one, None) == E_UNSUPPORTED
    d17 = L.ConvDesc(1, 64, 8, 8, 64, 8, 8, 1, 7, 1, 0, 3)                                  # 1xk / kx1 kernels (inception-resnet) have a weight-gradient plan
    assert lib.vcg_conv2d_wgrad_workspace_bytes(ctypes.byref(d17)) > lib.vcg_channel_sum_workspace_bytes(1, 64, 64)
    for code in (E_NULL, E_SHAPE, E_UNSUPPORTED, E_WORKSPACE):
        assert lib.vcg_error_string(code) and lib.vcg_error_string(code) != lib.vcg_error_string(0)
    # and the Python shim turns them into exceptions
    with pytest.raises((ValueError, RuntimeError)):
        L.check(E_SHAPE, "test")


def test_conv2d_bf16_constructor_only_accepts_trainable_shapes():
    """a Conv2DBf16 is built only for shapes its weight-gradient kernel serves (3x3 / 4x4, stride 1 / 2, channel
    multiples of 64) -- a 5x5 or 32-channel layer must fail at construction, not at the first training step"""
    from upscaler import _engine as E
    E.Conv2DBf16("ok", 64, 128, 4, 2, 1)
    for bad in ((32, 64, 3, 1), (64, 96, 3, 1), (64, 64, 5, 1), (64, 64, 3, 3)):
        with pytest.raises(NotImplementedError):
            E.Conv2DBf16("bad", bad[0], bad[1], bad[2], bad[3], "same")


def test_fused_step_is_opt_in_and_oracle_restates_it():
    """the faithful three-call step is the default everywhere; `fused_step` is an extension keyword behind the reference's arguments"""
    import inspect
    from oracle import train as OT
    from upscaler import model as PM
    for f in (PM.make_and_compile_gan, PM.make_and_compile_gan2):
        ps = inspect.signature(f).parameters
        assert ps["fused_step"].default is False and list(ps)[-1] == "fused_step"
    assert hasattr(OT.GanOracle, "train_step_fused")
    src = inspect.getsource(PM.GanTrainer.__init__)
    assert "self.fused = False" in src
