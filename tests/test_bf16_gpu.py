"""GPU parity of the bf16-storage kernels (BASELINE.json configs C3-C5) against the CPU oracle.

The oracle is evaluated in fp64 on the SAME bf16-rounded inputs and weights, so the only differences left are the
fp32 accumulation order and the final rounding of the output to bf16 (half an ulp = 2^-9 relative).  Tolerance,
max-norm relative: 2^-8 = 3.9e-3 (north_star's 1e-3 is stated for fp32; bf16 storage cannot resolve it)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err, report

pytestmark = pytest.mark.gpu
TOL_BF16 = 2.0 ** -8


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _to_nhwc_bf16(rt, x_nchw_f32_dev):
    from upscaler import _lib as L
    n, c, h, w = x_nchw_f32_dev.shape
    y = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_f32_nchw_to_bf16_nhwc(x_nchw_f32_dev.data_ptr(), y.data_ptr(), n, c, h, w, rt.stream), "to_bf16")
    return y


def _to_nchw_f32(rt, y_nhwc_bf16):
    from upscaler import _lib as L
    n, h, w, c = y_nhwc_bf16.shape
    o = torch.empty(n, c, h, w, dtype=torch.float32, device=rt.device)
    L.check(rt.lib.vcg_bf16_nhwc_to_f32_nchw(y_nhwc_bf16.data_ptr(), o.data_ptr(), n, c, h, w, rt.stream), "from_bf16")
    return o


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 7, 9), (1, 3, 5, 70), (2, 256, 4, 33), (1, 72, 3, 3)])
def test_layout_conversion_is_torch_rounding(rt, n, c, h, w):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g)
    xd = x.to(rt.device)
    y = _to_nhwc_bf16(rt, xd)
    ref = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    assert torch.equal(y.cpu().view(torch.int16), ref.view(torch.int16))        # round-to-nearest-even, bit-exact
    back = _to_nchw_f32(rt, y)
    assert torch.equal(back.cpu(), ref.float().permute(0, 3, 1, 2).contiguous())


@pytest.mark.parametrize("transpose,flip", [(1, 0), (0, 1), (0, 0), (1, 1)])
def test_pack_kernel(rt, transpose, flip):
    from upscaler import _lib as L
    taps, a, b = 9, 64, 48
    g = torch.Generator().manual_seed(4)
    w = torch.randn(taps, b, a, generator=g) if transpose else torch.randn(taps, a, b, generator=g)
    wd = w.to(rt.device)
    out = torch.empty(taps, a, b, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_pack_conv_kernel_bf16(wd.data_ptr(), taps, a, b, transpose, flip, out.data_ptr(), rt.stream), "pack")
    ref = w.transpose(1, 2) if transpose else w
    if flip:
        ref = ref.flip(0)
    assert torch.equal(out.cpu().view(torch.int16), ref.contiguous().to(torch.bfloat16).view(torch.int16))


CASES = [
    # n, h, w, scale/shift, act, residual
    (2, 16, 32, False, "none", False),
    (1, 13, 45, True, "prelu", False),        # ragged tile edges
    (3, 40, 72, True, "none", True),          # several tiles per image, BN-folded + Add
    (1, 64, 64, False, "lrelu", True),
    (2, 5, 7, True, "prelu", True),           # smaller than one tile
    (9, 33, 31, True, "prelu", True),         # more tiles than one wave of workgroups can hold per CU
    (16, 81, 97, True, "none", True),         # 384 tiles of 16x32: persistent workgroups take a second tile (double-buffered halo)
    (10, 100, 100, False, "lrelu", False),    # 280 tiles: some workgroups take two, most one
    (3, 200, 300, True, "prelu", False),      # 390 tiles, ragged on both edges
]


@pytest.mark.parametrize("n,h,w,affine,act,residual", CASES)
def test_conv3x3_c64_bf16(rt, n, h, w, affine, act, residual):
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    x = torch.randn(n, 64, h, w, generator=g)
    wk = torch.randn(3, 3, 64, 64, generator=g) * 0.06            # Keras (kh,kw,in,out)
    scale = (torch.rand(64, generator=g) + 0.5) if affine else None
    shift = (torch.rand(64, generator=g) - 0.5) if affine else None
    alpha = torch.rand(64, generator=g) * 0.5
    res = torch.randn(n, 64, h, w, generator=g) if residual else None

    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    wd = wk.to(rt.device)
    wp = torch.empty(9, 64, 64, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_pack_conv_kernel_bf16(wd.data_ptr(), 9, 64, 64, 1, 0, wp.data_ptr(), rt.stream), "pack")
    rd = _to_nhwc_bf16(rt, res.to(rt.device)) if residual else None
    sd = scale.to(rt.device) if affine else None
    hd = shift.to(rt.device) if affine else None
    ad = alpha.to(rt.device)
    y = torch.empty(n, h, w, 64, dtype=torch.bfloat16, device=rt.device)
    d = L.ConvDesc(n, 64, h, w, 64, h, w, 3, 3, 1, 1, 1)
    ep = L.EpilogueBf16(sd.data_ptr() if affine else None, hd.data_ptr() if affine else None,
                        {"none": L.ACT_NONE, "prelu": L.ACT_PRELU, "lrelu": L.ACT_LRELU}[act], 0.2,
                        ad.data_ptr() if act == "prelu" else None, rd.data_ptr() if residual else None)
    L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), xd.data_ptr(), wp.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream),
            "vcg_conv2d_bf16_fwd")
    got = _to_nchw_f32(rt, y).cpu().double()

    ref = K.conv2d(_bf16_round(x), _bf16_round(wk), None, 1, "same")
    if affine:
        ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == "prelu":
        ref = torch.clamp(ref, min=0) + alpha.double().view(1, -1, 1, 1) * torch.clamp(ref, max=0)
    elif act == "lrelu":
        ref = torch.where(ref >= 0, ref, 0.2 * ref)
    if residual:
        ref = ref + _bf16_round(res)
    e = rel_err(got, ref)
    # element-wise: half a bf16 ulp of the value + accumulation noise
    ew = float(((got - ref).abs() / (ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max())).max())
    report("bf16 conv3x3 c64 n=%d %dx%d affine=%s act=%s res=%s  err=%.2e  elementwise(ulp-scaled)=%.2f" % (n, h, w, affine, act, residual, e, ew))
    assert e < TOL_BF16
    assert ew < 1.0


CT_CASES = [
    # n, h, w, cout, lrelu
    (2, 12, 32, 64, False),
    (1, 13, 45, 128, True),         # ragged tile edges
    (2, 40, 72, 256, True),         # upsampling_block(64 -> 256)
    (1, 5, 7, 256, True),
    (4, 64, 96, 256, True),         # enough tiles for the LDS-tiled generic kernel (each phase: 4 x 8 x 3 tiles x 2 groups)
]


@pytest.mark.parametrize("n,h,w,cout,lrelu", CT_CASES)
def test_conv_transpose3x3_s2_bf16(rt, n, h, w, cout, lrelu):
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w + cout)
    x = torch.randn(n, 64, h, w, generator=g)
    wk = torch.randn(3, 3, cout, 64, generator=g) * 0.06          # Keras Conv2DTranspose (kh,kw,out,in)
    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    wd = wk.to(rt.device)
    wp = torch.empty(9, cout, 64, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_pack_conv_kernel_bf16(wd.data_ptr(), 9, cout, 64, 0, 0, wp.data_ptr(), rt.stream), "pack")
    y = torch.empty(n, 2 * h, 2 * w, cout, dtype=torch.bfloat16, device=rt.device)
    d = L.ConvDesc(n, 64, h, w, cout, 2 * h, 2 * w, 3, 3, 2, 0, 0)
    bias = torch.randn(cout, generator=g) * 0.3
    bd = bias.to(rt.device)
    ep = L.EpilogueBf16(None, bd.data_ptr(), L.ACT_LRELU if lrelu else L.ACT_NONE, 0.2, None, None)
    L.check(rt.lib.vcg_conv_transpose2d_bf16_fwd(ctypes.byref(d), xd.data_ptr(), wp.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream),
            "vcg_conv_transpose2d_bf16_fwd")
    got = _to_nchw_f32(rt, y).cpu().double()
    ref = K.conv2d_transpose_same(_bf16_round(x), _bf16_round(wk), bias.double(), 2)
    if lrelu:
        ref = torch.where(ref >= 0, ref, 0.2 * ref)
    e = rel_err(got, ref)
    ew = float(((got - ref).abs() / (ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max())).max())
    report("bf16 convT3x3 s2 64->%d n=%d %dx%d lrelu=%s  err=%.2e  elementwise(ulp-scaled)=%.2f" % (cout, n, h, w, lrelu, e, ew))
    assert e < TOL_BF16
    assert ew < 1.0
    # the same layer as the data gradient of the stride-2 convolution its kernel is (four phase launches of the generic kernels)
    wfr = torch.empty(9 * cout * 64, dtype=torch.bfloat16, device=rt.device)
    L.check(rt.lib.vcg_pack_conv_frag_bf16(wd.data_ptr(), 9, cout, 64, 1, wfr.data_ptr(), rt.stream), "pack frag")
    y2 = torch.full_like(y, float("nan"))
    L.check(rt.lib.vcg_conv_transpose2d_nhwc_bf16_fwd(ctypes.byref(d), xd.data_ptr(), wfr.data_ptr(), bd.data_ptr(), L.ACT_LRELU if lrelu else L.ACT_NONE, 0.2,
                                                      y2.data_ptr(), rt.stream), "vcg_conv_transpose2d_nhwc_bf16_fwd")
    got2 = _to_nchw_f32(rt, y2).cpu().double()
    e2 = rel_err(got2, ref)
    ew2 = float(((got2 - ref).abs() / (ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max())).max())
    report("  .. generic (phases): err=%.2e  elementwise(ulp-scaled)=%.2f" % (e2, ew2))
    assert e2 < TOL_BF16 and ew2 < 1.0


F9_CASES = [
    # n, h, w, tanh
    (1, 20, 64, True),
    (2, 37, 70, True),            # two strips (ragged), rows not a multiple of anything
    (1, 140, 40, False),          # several row segments
    (1, 6, 9, True),
]


@pytest.mark.parametrize("n,h,w,tanh", F9_CASES)
def test_final_conv9x9_256to3_bf16(rt, n, h, w, tanh):
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    x = torch.randn(n, 256, h, w, generator=g)
    wk = torch.randn(9, 9, 256, 3, generator=g) * 0.01
    b = torch.randn(3, generator=g) * 0.1
    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    wd, bd = wk.to(rt.device), b.to(rt.device)
    wf = torch.empty(L.FINAL9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
    L.check(rt.lib.vcg_pack_final9x9_bf16(wd.data_ptr(), wf.data_ptr(), rt.stream), "pack9")
    y = torch.empty(n, 3, h, w, dtype=torch.float32, device=rt.device)
    d = L.ConvDesc(n, 256, h, w, 3, h, w, 9, 9, 1, 4, 4)
    L.check(rt.lib.vcg_conv9x9_to3_bf16_fwd(ctypes.byref(d), xd.data_ptr(), wf.data_ptr(), bd.data_ptr(), 1 if tanh else 0, y.data_ptr(), rt.stream),
            "vcg_conv9x9_to3_bf16_fwd")
    ref = K.conv2d(_bf16_round(x), _bf16_round(wk), b.double(), 1, "same")
    if tanh:
        ref = torch.tanh(ref)
    e = rel_err(y.cpu().double(), ref)
    report("bf16 final conv9x9 256->3 n=%d %dx%d tanh=%s  err=%.2e (fp32 output)" % (n, h, w, tanh, e))
    assert e < 1e-4          # bf16 operands are exact in both; only the fp32 accumulation order differs


def _randomize_bn(G, seed):
    """non-trivial BatchNormalization statistics / affine parameters / PReLU slopes, as after training"""
    rng = np.random.RandomState(seed)
    w = G.get_weights_dict()
    for k, v in w.items():
        if k.endswith("/gamma"):
            w[k] = rng.uniform(0.7, 1.3, v.shape).astype(np.float32)
        elif k.endswith(("/beta", "/moving_mean", "/bias")):
            w[k] = rng.uniform(-0.2, 0.2, v.shape).astype(np.float32)
        elif k.endswith("/moving_variance"):
            w[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith("/alpha"):
            w[k] = rng.uniform(0.0, 0.3, v.shape).astype(np.float32)
    G.set_weights_dict(w)
    return w


@pytest.mark.parametrize("res,n,h,w", [(2, 2, 24, 40), (9, 1, 32, 32)])
def test_bf16_generator_matches_oracle_predict(rt, res, n, h, w):
    """C5: the whole inference pass (BN folded, bf16 storage, hipGraph replay) against the fp64 oracle's predict.
    bf16 storage rounds every activation tensor (2^-9 relative each, 2*res+3 tensors deep), so the end-to-end bound is
    looser than the per-kernel one: 3e-2 of the output range, with the fp32 product's 1e-3 beside it."""
    from oracle import models as M
    from upscaler import model as PM
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=res, seed=7)
    wd = _randomize_bn(G, 3)
    x = (np.random.RandomState(1).randint(0, 256, (n, h, w, 3)) / 127.5 - 1).astype(np.float32)
    ow = M.to_torch(wd, torch.float64)
    with torch.no_grad():
        ref, _ = M.upscaler_orig_forward(ow, torch.tensor(x, dtype=torch.float64), False, res, 2)        # NHWC in / out
    ref = ref.numpy()
    inf = G.to_inference_bf16()
    got = inf.predict(x)
    got2 = inf.predict(x)                                   # second call: pure graph replay
    assert np.array_equal(got, got2)
    e_bf16 = rel_err(got, ref)
    e_fp32 = rel_err(G.predict(x), ref)
    report("bf16 generator predict res=%d n=%d %dx%d  err=%.2e (fp32 product path %.2e)" % (res, n, h, w, e_bf16, e_fp32))
    assert e_fp32 < 1e-3
    assert e_bf16 < 3e-2
    # image-level: the uint8 frames differ by at most a few grey levels
    u_ref = np.around((ref + 1) * 127.5)
    u_got = np.around((got.astype(np.float64) + 1) * 127.5)
    report("bf16 generator uint8 frame difference: max %d levels, mean %.3f" % (np.abs(u_ref - u_got).max(), np.abs(u_ref - u_got).mean()))
    assert np.abs(u_ref - u_got).max() <= 6


@pytest.mark.parametrize("n,h,w,prelu", [(2, 12, 32, True), (1, 13, 45, True), (2, 40, 72, False), (1, 5, 7, True)])
def test_first_conv9x9_3to64_bf16(rt, n, h, w, prelu):
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    x = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    wk = torch.randn(9, 9, 3, 64, generator=g) * 0.1
    b = torch.randn(64, generator=g) * 0.2
    al = torch.rand(64, generator=g) * 0.5
    xd, wd, bd, ad = (t.to(rt.device) for t in (x, wk, b, al))
    wf = torch.empty(L.FIRST9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
    L.check(rt.lib.vcg_pack_first9x9_bf16(wd.data_ptr(), wf.data_ptr(), rt.stream), "pack_first")
    y = torch.empty(n, h, w, 64, dtype=torch.bfloat16, device=rt.device)
    d = L.ConvDesc(n, 3, h, w, 64, h, w, 9, 9, 1, 4, 4)
    L.check(rt.lib.vcg_conv9x9_from3_bf16_fwd(ctypes.byref(d), xd.data_ptr(), wf.data_ptr(), bd.data_ptr(), ad.data_ptr() if prelu else None,
                                              y.data_ptr(), rt.stream), "vcg_conv9x9_from3_bf16_fwd")
    got = _to_nchw_f32(rt, y).cpu().double()
    ref = K.conv2d(_bf16_round(x), _bf16_round(wk), b.double(), 1, "same")
    if prelu:
        ref = torch.clamp(ref, min=0) + al.double().view(1, -1, 1, 1) * torch.clamp(ref, max=0)
    e = rel_err(got, ref)
    ew = float(((got - ref).abs() / (ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max())).max())
    report("bf16 first conv9x9 3->64 n=%d %dx%d prelu=%s  err=%.2e  elementwise(ulp-scaled)=%.2f" % (n, h, w, prelu, e, ew))
    assert e < TOL_BF16 and ew < 1.0


@pytest.mark.parametrize("mode,n,c,h,w,act,residual", [("batch", 3, 64, 9, 13, "prelu", True), ("instance", 2, 64, 50, 47, "none", True),
                                                      ("instance", 2, 128, 7, 5, "lrelu", False), ("batch", 2, 256, 40, 60, "prelu", False)])
def test_norm_bf16(rt, mode, n, c, h, w, act, residual):
    """bf16 NHWC statistics (+ the fp32 finalize) and normalise + activation + Add against fp64 on the same bf16 values"""
    from upscaler import _engine as E, _lib as L
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g) * 1.7 + torch.randn(1, c, 1, 1, generator=g) * 3.0      # |mean| ~ std: the shift matters
    res = torch.randn(n, c, h, w, generator=g) if residual else None
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    alpha = torch.rand(c, generator=g) * 0.4
    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    rows = n if mode == "instance" else 1
    code = L.NORM_INSTANCE if mode == "instance" else L.NORM_BATCH
    mean, var, scale, shift, invstd = (torch.empty(rows * c, device=rt.device) for _ in range(5))
    ws, wsn = rt.workspace(rt.lib.vcg_norm_stats_bf16_workspace_bytes(n, c, h * w, code))
    L.check(rt.lib.vcg_norm_stats_bf16(xd.data_ptr(), n, c, h * w, code, mean.data_ptr(), var.data_ptr(), ws, wsn, rt.stream), "stats")
    xb = _bf16_round(x)
    dims = (2, 3) if mode == "instance" else (0, 2, 3)
    mref, vref = xb.mean(dims), xb.var(dims, unbiased=False)
    e_m, e_v = rel_err(mean.cpu().view(mref.shape), mref), rel_err(var.cpu().view(vref.shape), vref)
    gd, bd, ad = gamma.to(rt.device), beta.to(rt.device), alpha.to(rt.device)
    eps = E.IN_EPS if mode == "instance" else E.BN_EPS
    L.check(rt.lib.vcg_norm_finalize(mean.data_ptr(), var.data_ptr(), None if mode == "instance" else gd.data_ptr(),
                                     None if mode == "instance" else bd.data_ptr(), c, rows, eps, scale.data_ptr(), shift.data_ptr(),
                                     invstd.data_ptr(), None, None, 0.0, 0, rt.stream), "finalize")
    rd = _to_nhwc_bf16(rt, res.to(rt.device)) if residual else None
    y = torch.empty_like(xd)
    L.check(rt.lib.vcg_norm_act_fwd_bf16(xd.data_ptr(), n, c, h * w, scale.data_ptr(), shift.data_ptr(), 1 if mode == "instance" else 0,
                                         {"none": L.ACT_NONE, "prelu": L.ACT_PRELU, "lrelu": L.ACT_LRELU}[act], 0.2,
                                         ad.data_ptr() if act == "prelu" else None, rd.data_ptr() if residual else None, y.data_ptr(),
                                         rt.stream), "norm_act")
    got = _to_nchw_f32(rt, y).cpu().double()
    shp = (n, c, 1, 1) if mode == "instance" else (1, c, 1, 1)
    ref = (xb - mref.view(shp)) / torch.sqrt(vref.view(shp) + eps)
    if mode == "batch":
        ref = ref * gamma.double().view(1, c, 1, 1) + beta.double().view(1, c, 1, 1)
    if act == "prelu":
        ref = torch.clamp(ref, min=0) + alpha.double().view(1, c, 1, 1) * torch.clamp(ref, max=0)
    elif act == "lrelu":
        ref = torch.where(ref > 0, ref, 0.2 * ref)
    if residual:
        ref = ref + _bf16_round(res)
    e = rel_err(got, ref)
    report("bf16 norm %s n=%d c=%d %dx%d act=%s res=%s  mean err=%.1e var err=%.1e out err=%.2e" % (mode, n, c, h, w, act, residual, e_m, e_v, e))
    assert e_m < 1e-5 and e_v < 1e-4 and e < TOL_BF16          # fp32 sums of squares over up to 1e5 bf16 samples


def test_bf16_generator_instance_norm(rt):
    """norm='instance' generator (north_star's instance-norm variant) through the bf16 engine: per-image statistics on bf16"""
    from oracle import models as M
    from upscaler import model as PM
    n, h, w, res = 2, 24, 40, 2
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=res, norm="instance", seed=7)
    wd = _randomize_bn(G, 3)
    x = (np.random.RandomState(1).randint(0, 256, (n, h, w, 3)) / 127.5 - 1).astype(np.float32)
    with torch.no_grad():
        ref, _ = M.upscaler_orig_forward(M.to_torch(wd, torch.float64), torch.tensor(x, dtype=torch.float64), False, res, 2, norm="instance")
    got = G.to_inference_bf16().predict(x)
    e = rel_err(got, ref.numpy())
    report("bf16 generator (instance norm) predict res=%d n=%d %dx%d  err=%.2e" % (res, n, h, w, e))
    assert e < 3e-2                                 # observed 1.3e-2 (per-image statistics are re-derived from bf16-stored tensors)


@pytest.mark.parametrize("n,h,w", [(1, 8, 32), (2, 16, 64), (1, 13, 45), (3, 40, 72), (2, 5, 7)])
def test_conv3x3_c64_bf16_wgrad(rt, n, h, w):
    """weight / bias gradient of the bf16 trunk convolution (transposed LDS reads) against fp64 autograd on the same
    bf16-rounded operands; the result is fp32, so only the accumulation order differs: 1e-4"""
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    x = torch.randn(n, 64, h, w, generator=g)
    dy = torch.randn(n, 64, h, w, generator=g)
    xb, dyb = _bf16_round(x), _bf16_round(dy)
    wk = torch.zeros(3, 3, 64, 64, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(64, dtype=torch.float64, requires_grad=True)
    (K.conv2d(xb, wk, b, 1, "same") * dyb).sum().backward()
    xd, dyd = _to_nhwc_bf16(rt, x.to(rt.device)), _to_nhwc_bf16(rt, dy.to(rt.device))
    dw = torch.empty(3, 3, 64, 64, dtype=torch.float32, device=rt.device)
    db = torch.empty(64, dtype=torch.float32, device=rt.device)
    d = L.ConvDesc(n, 64, h, w, 64, h, w, 3, 3, 1, 1, 1)
    ws, wsn = rt.workspace(rt.lib.vcg_conv2d_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
    L.check(rt.lib.vcg_conv2d_bf16_wgrad(ctypes.byref(d), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws, wsn, rt.stream),
            "vcg_conv2d_bf16_wgrad")
    e_w, e_b = rel_err(dw, wk.grad), rel_err(db, b.grad)
    report("bf16 wgrad 3x3 c64 n=%d %dx%d  dw err=%.2e  db err=%.2e" % (n, h, w, e_w, e_b))
    assert e_w < 1e-4 and e_b < 1e-4


@pytest.mark.parametrize("mode,n,c,h,w,act", [("batch", 4, 64, 24, 20, "prelu"), ("instance", 2, 64, 30, 33, "none"),
                                              ("batch", 2, 128, 9, 11, "lrelu"), ("instance", 3, 64, 16, 16, "prelu")])
def test_norm_bwd_bf16(rt, mode, n, c, h, w, act):
    """bf16 NHWC norm + activation backward against fp64 autograd on the same bf16 values: dx (bf16 output, 2^-8), and
    the fp32 parameter gradients dgamma / dbeta / dalpha (1e-4)"""
    from upscaler import _engine as E, _lib as L
    g = torch.Generator().manual_seed(c + h + n)
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + torch.randn(1, c, 1, 1, generator=g)
    dy = torch.randn(n, c, h, w, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    alpha = torch.rand(c, generator=g) * 0.4 + 0.05
    inst = mode == "instance"
    eps = E.IN_EPS if inst else E.BN_EPS
    xb, dyb = _bf16_round(x).requires_grad_(True), _bf16_round(dy)
    gr, br, ar = (t.double().clone().requires_grad_(True) for t in (gamma, beta, alpha))
    dims = (2, 3) if inst else (0, 2, 3)
    mu, var = xb.mean(dims, keepdim=True), xb.var(dims, unbiased=False, keepdim=True)
    xh = (xb - mu) / torch.sqrt(var + eps)
    u = xh if inst else xh * gr.view(1, c, 1, 1) + br.view(1, c, 1, 1)
    if act == "prelu":
        y = torch.clamp(u, min=0) + ar.view(1, c, 1, 1) * torch.clamp(u, max=0)
    elif act == "lrelu":
        y = torch.where(u > 0, u, 0.2 * u)
    else:
        y = u
    (y * dyb).sum().backward()

    xd, dyd = _to_nhwc_bf16(rt, x.to(rt.device)), _to_nhwc_bf16(rt, dy.to(rt.device))
    rows = n if inst else 1
    code = L.NORM_INSTANCE if inst else L.NORM_BATCH
    mean, varr, scale, shift, invstd = (torch.empty(rows * c, device=rt.device) for _ in range(5))
    ws, wsn = rt.workspace(rt.lib.vcg_norm_stats_bf16_workspace_bytes(n, c, h * w, code))
    L.check(rt.lib.vcg_norm_stats_bf16(xd.data_ptr(), n, c, h * w, code, mean.data_ptr(), varr.data_ptr(), ws, wsn, rt.stream), "stats")
    gd, bd, ad = gamma.to(rt.device), beta.to(rt.device), alpha.to(rt.device)
    L.check(rt.lib.vcg_norm_finalize(mean.data_ptr(), varr.data_ptr(), None if inst else gd.data_ptr(), None if inst else bd.data_ptr(), c, rows,
                                     eps, scale.data_ptr(), shift.data_ptr(), invstd.data_ptr(), None, None, 0.0, 0, rt.stream), "finalize")
    dx = torch.empty_like(xd)
    dga, dbe, dal = (torch.zeros(c, device=rt.device) for _ in range(3))
    ws, wsn = rt.workspace(rt.lib.vcg_norm_act_bwd_bf16_workspace_bytes(n, c, h * w, code))
    L.check(rt.lib.vcg_norm_act_bwd_bf16(xd.data_ptr(), dyd.data_ptr(), n, c, h * w, code, mean.data_ptr(), invstd.data_ptr(),
                                         None if inst else gd.data_ptr(), None if inst else bd.data_ptr(),
                                         {"none": L.ACT_NONE, "prelu": L.ACT_PRELU, "lrelu": L.ACT_LRELU}[act], 0.2,
                                         ad.data_ptr() if act == "prelu" else None, 1, dx.data_ptr(), None if inst else dga.data_ptr(),
                                         None if inst else dbe.data_ptr(), dal.data_ptr() if act == "prelu" else None, ws, wsn, rt.stream), "norm_bwd")
    e_dx = rel_err(_to_nchw_f32(rt, dx), xb.grad)
    errs = {"dx": e_dx}
    if not inst:
        errs["dgamma"], errs["dbeta"] = rel_err(dga, gr.grad), rel_err(dbe, br.grad)
    if act == "prelu":
        errs["dalpha"] = rel_err(dal, ar.grad)
    report("bf16 norm bwd %s n=%d c=%d %dx%d act=%s  %s" % (mode, n, c, h, w, act, "  ".join("%s=%.2e" % kv for kv in errs.items())))
    assert e_dx < TOL_BF16
    assert all(v < 2e-4 for k, v in errs.items() if k != "dx")


@pytest.mark.parametrize("mode", ["bf16", "bf16+tail"])
def test_bf16_trunk_generator_training_forward_and_gradients(rt, mode):
    """make_upscaler_orig(..., trunk_dtype='bf16'): the residual trunk trains on bf16 activations (conv fwd / dgrad /
    wgrad, norm fwd / bwd in bf16_*.hip) with fp32 master weights.  Training-mode forward, loss and every gradient tensor
    against the fp64 oracle evaluated WITH THE SAME STORAGE ROUNDINGS (oracle.keras_ops.bf16_store at the tensors the
    product keeps in bf16, values and gradients): what remains is fp32-vs-fp64 arithmetic, measured by running the same
    emulation in fp32 (bound: 2.5x that distance, or the fp32 path's 1e-3 / 1e-2 if larger).  The distance to the un-rounded
    fp64 oracle -- the price of bf16 storage itself -- is reported beside it."""
    from oracle import models as M
    from upscaler import model as PM, _engine as E
    res, n, h, w = 2, 4, 32, 32
    Gb = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=res, seed=7, trunk_dtype=mode)
    Gf = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=res, seed=7)
    wd = _randomize_bn(Gb, 5)
    Gf.set_weights_dict(wd)
    x = (np.random.RandomState(1).randint(0, 256, (n, h, w, 3)) / 127.5 - 1).astype(np.float32)
    t = (np.random.RandomState(2).randint(0, 256, (n, 2 * h, 2 * w, 3)) / 127.5 - 1).astype(np.float32)

    def oracle(trunk_bf16, dt=torch.float64):
        leaf = M.to_torch(wd, dt, requires_grad=True)
        yr, upd = M.upscaler_orig_forward(leaf, torch.tensor(x, dtype=dt), True, res, 2, trunk_bf16=trunk_bf16,
                                          tail_bf16=trunk_bf16 and mode == "bf16+tail")
        loss = ((yr - torch.tensor(t, dtype=dt)) ** 2).mean()
        names = [k for k, v in leaf.items() if v.requires_grad]
        return yr.detach().double(), float(loss.detach()), dict(zip(names, [g.double() for g in torch.autograd.grad(loss, [leaf[k] for k in names])])), upd
    ref = {"bf16-trunk": oracle(True), "fp32": oracle(False)}
    # the yardstick: the SAME emulation evaluated in fp32.  With bf16 storage, fp32-vs-fp64 arithmetic no longer means 1e-6:
    # a value within fp32 rounding of a bf16 boundary is stored as the other neighbour (one bf16 ulp), and the trunk amplifies
    # these flips (measured: output 3e-3, gradient tensors up to 3.4e-2 in relative L2 between the two oracle runs)
    y32, _, g32, _ = oracle(True, torch.float32)
    out = {}
    for tag, G in (("bf16-trunk", Gb), ("fp32", Gf)):
        y, tape = G.forward(E.to_device_nchw(rt, x), True)
        val, dy = PM._pixel_loss(rt, y, E.to_device_nchw(rt, t), "mse", 1.0)
        G.backward(tape, dy, 0)
        yr, lossr, gref, upd = ref[tag]
        l2 = lambda a, b, floor=0.0: float((a - b).norm() / (b.norm() + floor))
        yd = E.to_nhwc(rt, y).cpu().double()
        e_y = l2(yd, yr)          # relative L2: a value within fp32 rounding of a bf16 rounding boundary lands on the other
        #                           neighbour (one bf16 ulp, 4e-3 of that element) -- a handful of such elements set the max-norm
        gmax = max(float(g.abs().max()) for g in gref.values())
        worst, worst_plain, worst_ratio = 0.0, 0.0, 0.0
        for k, b in gref.items():
            a = G.ps.grad(k).cpu().double()
            floor = 1e-4 * gmax * b.numel() ** 0.5
            # tensors whose gradient is (numerically) zero -- conv biases in front of a BatchNormalization -- carry rounding
            # noise only: they are excluded from the bound (their error is measured against the floor and reported)
            real = float(b.norm()) >= floor
            e = l2(a, b, floor)
            worst_plain = max(worst_plain, l2(a, ref["fp32"][2][k], floor))
            if tag == "bf16-trunk":
                # PER-TENSOR yardstick: this tensor's own distance between the fp32 and fp64 runs of the same emulation
                e32 = l2(g32[k], b, floor)
                report("    %-40s |g|2=%.2e rel L2 err=%.2e (oracle fp32-vs-fp64, same storage: %.2e)%s"
                       % (k, float(b.norm()), e, e32, "" if real else "   [zero gradient: excluded]"))
                if real:
                    worst_ratio = max(worst_ratio, e / max(e32, 4e-3))
                    assert e < max(1e-2, 2.5 * e32), (k, e, e32)
            if real:
                worst = max(worst, e)
        out[tag] = (e_y, worst, abs(float(val.item()) - lossr) / lossr)
        if tag == "bf16-trunk":
            e32_y = l2(y32, yr)
            report("    oracle fp32-with-bf16-storage vs oracle fp64-with-bf16-storage: output %.2e; worst product/oracle-fp32 ratio over real gradient tensors %.2f"
                   % (e32_y, worst_ratio))
            assert e_y < max(1e-3, 2.5 * e32_y) and out[tag][2] < 1e-4, (out[tag], e32_y)
        report("generator training pass [" + mode + "] (%s) vs oracle with the same storage: output err (rel L2)=%.2e  worst gradient tensor (rel L2)=%.2e  loss err=%.1e"
               "   [vs un-rounded fp64 oracle: output %.2e, gradients %.2e]"
               % ((tag,) + out[tag] + (l2(yd, ref["fp32"][0]), worst_plain)))
        sw = G.get_weights_dict()
        for k, v in upd.items():                          # moving statistics: momentum 0.99, Bessel-corrected variance
            assert np.max(np.abs(sw[k] - v.detach().numpy())) < 1e-4 * (np.max(np.abs(v.detach().numpy())) + 1e-3), k
    assert out["fp32"][0] < 1e-3 and out["fp32"][1] < 1e-2 and out["fp32"][2] < 1e-4, out["fp32"]


def test_bf16_generator_at_c4_frame_size(rt):
    """bf16 inference at BASELINE.json config C4's frame size (540x960 -> 1080x1920; tiles of 12x32 pixels, 64-column strips and
    row segments all end raggedly or exactly at the frame edge): a band of the frame against the oracle, which sees the same
    receptive field away from the band's edges"""
    from oracle import models as M
    from upscaler import model as PM
    h, w, res = 540, 960, 2
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=res, seed=7)
    wd = _randomize_bn(G, 3)
    x = (np.random.RandomState(4).randint(0, 256, (1, h, w, 3)) / 127.5 - 1).astype(np.float32)
    got = G.to_inference_bf16().predict(x)
    assert got.shape == (1, 2 * h, 2 * w, 3) and np.isfinite(got).all()
    for r0 in (0, 254, h - 32):                                  # top edge, interior, bottom edge
        band = x[:, r0:r0 + 32]
        with torch.no_grad():
            ref, _ = M.upscaler_orig_forward(M.to_torch(wd, torch.float64), torch.tensor(band, dtype=torch.float64), False, res, 2)
        lo = 0 if r0 == 0 else 14                                # rows whose receptive field stays inside the band (or hits the
        hi = 32 if r0 == h - 32 else 18                          # true frame edge, where both pad with zeros)
        e = rel_err(got[:, 2 * (r0 + lo):2 * (r0 + hi)], ref[:, 2 * lo:2 * hi].numpy())
        report("bf16 generator at 540x960->1080x1920, LR rows %d..%d: err=%.2e" % (r0 + lo, r0 + hi, e))
        assert e < 2e-2                             # observed 5.2e-3 .. 6.2e-3 (2 blocks + tail)


@pytest.mark.parametrize("n,h,w,mask", [(1, 12, 32, False), (2, 37, 70, True), (1, 5, 9, True)])
def test_final_conv9x9_bf16_dgrad(rt, n, h, w, mask):
    """data gradient of final/conv on the bf16 path (the 3-channel 9x9 kernel run with the taps flipped, 4 channel blocks), with
    the LeakyReLU backward of the tensor in front of it folded into the epilogue"""
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 100 + h + w)
    wk = torch.randn(9, 9, 256, 3, generator=g) * 0.02
    dy = torch.randn(n, 3, h, w, generator=g)
    yprev = torch.randn(n, 256, h, w, generator=g)
    x = torch.zeros(n, 256, h, w, dtype=torch.float64, requires_grad=True)
    (K.conv2d(x, _bf16_round(wk), None, 1, "same") * _bf16_round(dy)).sum().backward()
    ref = x.grad
    if mask:
        ref = ref * torch.where(_bf16_round(yprev) > 0, 1.0, 0.2)
    wf = torch.empty(4 * L.FIRST9X9_WFRAG_BYTES, dtype=torch.uint8, device=rt.device)
    wd = wk.to(rt.device)
    L.check(rt.lib.vcg_pack_conv9x9_3ch_bf16(wd.data_ptr(), 256, 1, wf.data_ptr(), rt.stream), "pack dgrad")
    dyd = dy.to(rt.device)
    yp = _to_nhwc_bf16(rt, yprev.to(rt.device))
    dx = torch.empty(n, h, w, 256, dtype=torch.bfloat16, device=rt.device)
    d = L.ConvDesc(n, 256, h, w, 3, h, w, 9, 9, 1, 4, 4)
    L.check(rt.lib.vcg_conv9x9_to3_bf16_dgrad(ctypes.byref(d), dyd.data_ptr(), wf.data_ptr(), yp.data_ptr() if mask else None, 0.2, dx.data_ptr(),
                                              rt.stream), "vcg_conv9x9_to3_bf16_dgrad")
    e = rel_err(_to_nchw_f32(rt, dx), ref)
    report("bf16 final conv dgrad 3->256 n=%d %dx%d mask=%s err=%.2e" % (n, h, w, mask, e))
    assert e < TOL_BF16


@pytest.mark.parametrize("n,h,w", [(1, 16, 32), (2, 37, 70), (1, 5, 10), (3, 64, 96)])
def test_final_conv9x9_bf16_wgrad(rt, n, h, w):
    """weight gradient of final/conv on the bf16 path: x bf16 NHWC by transposed LDS reads, dz as a bf16 operand (8 taps x (3 + 1)
    channels per MFMA column tile), against the fp64 oracle on the same rounded operands; deterministic"""
    from oracle import keras_ops as K
    from upscaler import _lib as L
    g = torch.Generator().manual_seed(n * 100 + h + w)
    x = torch.randn(n, 256, h, w, generator=g)
    dz = torch.randn(n, 3, h, w, generator=g)
    wk = torch.zeros(9, 9, 256, 3, dtype=torch.float64, requires_grad=True)
    (K.conv2d(_bf16_round(x), wk, None, 1, "same") * _bf16_round(dz)).sum().backward()
    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    dzd = dz.to(rt.device)
    d = L.ConvDesc(n, 256, h, w, 3, h, w, 9, 9, 1, 4, 4)
    ws, wsn = rt.workspace(rt.lib.vcg_conv9x9_to3_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
    outs = []
    for _ in range(2):
        dw = torch.full((9, 9, 256, 3), float("nan"), device=rt.device)
        L.check(rt.lib.vcg_conv9x9_to3_bf16_wgrad(ctypes.byref(d), xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), ws, wsn, rt.stream), "vcg_conv9x9_to3_bf16_wgrad")
        outs.append(dw.clone())
    e = rel_err(outs[0], wk.grad)
    report("bf16 final conv wgrad 256->3 n=%d %dx%d err=%.2e" % (n, h, w, e))
    assert e < 1e-5
    assert torch.equal(outs[0], outs[1])
    dodd = L.ConvDesc(n, 256, h, w + 1, 3, h, w + 1, 9, 9, 1, 4, 4)
    assert rt.lib.vcg_conv9x9_to3_bf16_wgrad(ctypes.byref(dodd), xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), ws, wsn, rt.stream) == -3      # VCG_E_UNSUPPORTED: odd width


@pytest.mark.parametrize("mode", ["bf16", "bf16+tail"])
def test_bf16_generator_modes_graph_replay_matches_eager(rt, mode):
    """the mixed-precision generator modes inside the captured train step: bf16 weight copies are re-packed inside the graph after
    every Adam update, so the replays reproduce the eagerly launched steps bit for bit"""
    from upscaler import model as PM, _engine as E

    def run(graph):
        G = PM.make_upscaler_orig((64, 64, 3), kernel_size=3, upscale_factor=2, res_block_num=2, seed=7, trunk_dtype=mode)
        D = PM.make_discriminator_patchgan_70((64, 64, 3), seed=11)
        _, _, gan = PM.make_and_compile_gan2(G, D, (32, 32, 3), (64, 64, 3), "mse", 1.0, lambda: PM.WassersteinLosses(), 1e-2,
                                             optimizer=PM.Adam())
        tr = gan.trainer
        rng = np.random.RandomState(5)
        steps = [(E.to_device_nchw(rt, rng.randint(0, 256, (2, 32, 32, 3)) / 127.5 - 1),
                  E.to_device_nchw(rt, rng.randint(0, 256, (2, 64, 64, 3)) / 127.5 - 1)) for _ in range(4)]
        out = []
        if graph:
            tr.capture_train_step(*steps[0])
            for a, b in steps[1:]:
                out.append(tr.train_step_graph(a, b))
        else:
            for a, b in steps:
                out.append(tr.train_step(a, b))
            out = out[1:]
        return out, G.ps.params.clone(), D.ps.params.clone()
    oe, ge, de = run(False)
    og, gg, dg = run(True)
    assert oe == og, (oe, og)
    assert torch.equal(ge, gg) and torch.equal(de, dg)
    assert all(np.isfinite(v) for step in og for v in step)


# ---------------------------------------------------------------------------------------------------------------
# generic bf16 NHWC convolution (the discriminators' layers in the bf16 configs): forward, data and weight gradient
# ---------------------------------------------------------------------------------------------------------------
GCONV_CASES = [
    # cin, cout, k, stride, padding, n, h, w
    (64, 128, 4, 2, 1, 2, 32, 32),             # PatchGAN block 2
    (128, 256, 4, 2, 1, 1, 16, 16),            # PatchGAN block 3
    (256, 512, 4, 1, 1, 1, 10, 10),            # PatchGAN block 4 (ragged: 9x9 output)
    (64, 128, 3, 2, "same", 2, 32, 32),        # simple_512 / thin_512 block 2 (TF-SAME: pads (0,1))
    (128, 128, 3, 2, "same", 1, 15, 17),       # odd sizes: pads (1,1), ragged tiles
    (512, 512, 3, 2, "same", 2, 4, 4),
    (512, 512, 3, 2, "same", 3, 2, 2),         # 2x2 -> 1x1
    (64, 64, 3, 1, "same", 1, 9, 20),
    (64, 128, 4, 2, 1, 3, 70, 38),             # many tiles, several images
    # large enough for the LDS-tiled kernel (unit input stride, >= 64 (tile, 128-channel group) pairs): forward of stride-1 layers,
    # data gradients of any stride
    (256, 512, 4, 1, 1, 2, 70, 50),            # PatchGAN block 4: forward (4x4 box of taps) and data gradient on the LDS kernel, ragged tiles
    (64, 128, 3, 1, "same", 4, 64, 64),        # forward on the LDS kernel (3x3 box)
    (128, 256, 4, 2, 1, 4, 128, 128),          # data gradient of a stride-2 layer: four phases of 2x2 taps
    (256, 256, 3, 2, "same", 8, 64, 61),       # 3x3 stride 2: phases / parity planes with one or two taps per dimension (absent offsets), ragged
    (64, 64, 3, 1, "same", 8, 64, 64),         # 64 output channels: a ragged 128-channel group (two of the four waves compute nothing)
    (64, 192, 3, 1, "same", 4, 64, 64),        # 192 = one full group + one ragged
    (64, 128, 4, 2, 1, 4, 128, 96),            # stride-2 forward through the four parity planes (PatchGAN block 2), 64-channel gradient groups
    (128, 128, 3, 1, "same", 5, 67, 45),       # odd sizes on the LDS kernel, two input chunks
    (128, 128, 3, 2, "same", 7, 75, 83),       # stride 2 on odd sizes: TF-SAME pads (1,1), parity planes of unequal extent
    (64, 128, 4, 1, 1, 8, 40, 50),             # 4x4 box with a 64-channel data gradient: the row-split form with loader waves, three stages of 52 KiB
    (128, 64, 4, 1, 1, 8, 40, 50),             # ... and a 64-channel forward in that form
    (64, 64, 3, 2, "same", 8, 66, 70),         # 64 -> 64 stride 2: that form on parity planes (forward) and phases (data gradient), ragged
]


@pytest.mark.parametrize("cin,cout,k,stride,padding,n,h,w", GCONV_CASES)
def test_generic_conv_bf16_fwd_dgrad_wgrad(rt, cin, cout, k, stride, padding, n, h, w):
    """Conv2DBf16 (vcg_conv2d_nhwc_bf16_fwd / _dgrad / _wgrad) against the fp64 oracle on the same bf16-rounded operands:
    outputs that are stored in bf16 to 2^-8 max-norm, the fp32 weight / bias gradients to 1e-4."""
    from oracle import keras_ops as K
    from upscaler import _engine as E
    layer = E.Conv2DBf16("c", cin, cout, k, stride, padding)
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    g = torch.Generator().manual_seed(cin + cout + k * 7 + h)
    wk = torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5
    bk = torch.randn(cout, generator=g) * 0.1
    ps.set_weights({"c/kernel": wk.numpy(), "c/bias": bk.numpy()})
    x = torch.randn(n, cin, h, w, generator=g)
    xr = _bf16_round(x).requires_grad_(True)
    wr = _bf16_round(wk).requires_grad_(True)
    br = bk.double().requires_grad_(True)
    yr = K.conv2d(xr, wr, br, stride, padding)
    dy = torch.randn(*yr.shape, generator=g)
    dyr = _bf16_round(dy)
    (yr * dyr).sum().backward()

    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    y, ctx = layer.forward(xd)
    assert tuple(y.shape) == (n, yr.shape[2], yr.shape[3], cout)
    dx = layer.backward(ctx, _to_nhwc_bf16(rt, dy.to(rt.device)), True, True, 0)
    e_y = rel_err(_to_nchw_f32(rt, y), yr)
    e_dx = rel_err(_to_nchw_f32(rt, dx), xr.grad)
    e_dw = rel_err(ps.grad("c/kernel"), wr.grad)
    e_db = rel_err(ps.grad("c/bias"), br.grad)
    report("generic bf16 conv %d->%d k%d s%d pad=%s n=%d %dx%d: fwd=%.2e dgrad=%.2e wgrad=%.2e dbias=%.2e" % (cin, cout, k, stride, padding, n, h, w, e_y, e_dx, e_dw, e_db))
    assert e_y < TOL_BF16 and e_dx < TOL_BF16
    assert e_dw < 1e-4 and e_db < 1e-4
    # deterministic: a second backward reproduces the gradients bit for bit
    g1 = ps.grad("c/kernel").clone()
    layer.backward(ctx, _to_nhwc_bf16(rt, dy.to(rt.device)), False, True, 0)
    assert torch.equal(g1, ps.grad("c/kernel"))


@pytest.mark.parametrize("cin,k,padding,n,h,w", [
    (512, 4, 1, 2, 15, 15),          # the PatchGAN head at a 64x64... input: 15 -> 14
    (512, 4, 1, 3, 63, 63),          # its size at C2 / C3 (512x512 frames): 63 -> 62, several segments per row, ragged last segment
    (512, 4, 1, 1, 20, 37),          # not square
    (256, 4, 1, 2, 9, 11),           # fewer than 512 channels: idle lanes
    (64, 3, "same", 2, 17, 16),      # 3x3 'same'
    (512, 3, 1, 1, 8, 33),
])
def test_cout1_conv_bf16_fwd_dgrad_wgrad(rt, cin, k, padding, n, h, w):
    """ConvCout1Bf16 (vcg_conv2d_cout1_nhwc_bf16_*: the PatchGAN head on bf16 NHWC activations, fp32 weights) against the fp64 oracle on
    the same bf16-rounded activations: fp32 outputs / weight gradients to 1e-5, the bf16-stored data gradient to 2^-8."""
    from oracle import keras_ops as K
    from upscaler import _engine as E
    layer = E.ConvCout1Bf16("c", cin, 1, k, 1, padding)
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    g = torch.Generator().manual_seed(cin + k * 7 + h)
    wk = torch.randn(k, k, cin, 1, generator=g) * (2.0 / (k * k * cin)) ** 0.5
    bk = torch.randn(1, generator=g) * 0.1
    ps.set_weights({"c/kernel": wk.numpy(), "c/bias": bk.numpy()})
    x = torch.randn(n, cin, h, w, generator=g)
    xr = _bf16_round(x).requires_grad_(True)
    wr = wk.double().requires_grad_(True)
    br = bk.double().requires_grad_(True)
    yr = K.conv2d(xr, wr, br, 1, padding)
    dy = torch.randn(*yr.shape, generator=g)
    (yr * dy.double()).sum().backward()
    xd = _to_nhwc_bf16(rt, x.to(rt.device))
    y, ctx = layer.forward(xd)
    assert tuple(y.shape) == tuple(yr.shape)
    dx = layer.backward(ctx, dy.to(rt.device).contiguous(), True, True, 0)
    e_y = rel_err(y, yr)
    e_dx = rel_err(_to_nchw_f32(rt, dx), xr.grad)
    e_dw = rel_err(ps.grad("c/kernel"), wr.grad)
    e_db = rel_err(ps.grad("c/bias"), br.grad)
    report("cout1 bf16 conv %d->1 k%d pad=%s n=%d %dx%d: fwd=%.2e dgrad=%.2e wgrad=%.2e dbias=%.2e" % (cin, k, padding, n, h, w, e_y, e_dx, e_dw, e_db))
    assert e_y < 1e-5 and e_dw < 1e-5 and e_db < 1e-5
    assert e_dx < TOL_BF16
    g1 = ps.grad("c/kernel").clone()
    layer.backward(ctx, dy.to(rt.device).contiguous(), False, True, 0)
    assert torch.equal(g1, ps.grad("c/kernel"))


@pytest.mark.parametrize("cout,k,stride,padding,slope,n,h,w", [
    (64, 4, 2, 1, 0.2, 2, 64, 64),           # PatchGAN block 1
    (64, 4, 2, 1, 0.2, 1, 70, 54),           # ragged tiles (35 x 27 outputs)
    (64, 4, 2, 1, 0.2, 3, 33, 47),           # odd sizes: the last row / column of taps falls into the padding
    (64, 3, 1, "same", None, 2, 40, 72),     # simple_512 / thin_512 block 1 (model.py:839): no activation (BatchNormalization follows)
    (128, 3, 1, "same", 0.1, 1, 25, 33),     # two output-channel blocks
])
def test_first_conv_bf16_forward_and_gradients(rt, cout, k, stride, padding, slope, n, h, w):
    """FirstConvBf16 (vcg_conv3ch_bf16_fwd: fp32 NCHW frames -> bf16 NHWC, + bias + LeakyReLU) against the fp64 oracle on the same
    bf16-rounded frames / kernel: output to 2^-8; weight / bias gradients (vcg_conv3ch_bf16_wgrad: the bf16 copy of the frames x the bf16
    gradient in front of the activation, fp32 accumulation) to 1e-5 against autograd with the straight-through gradient of the roundings; the data
    gradient (vcg_conv3ch_bf16_dgrad: bf16 kernel copy, virtual-channel convolution stored in bf16) to 2^-8."""
    from oracle import keras_ops as K
    from upscaler import _engine as E, _lib as L
    layer = E.FirstConvBf16("c", 3, cout, k, stride, padding, L.ACT_LRELU if slope else L.ACT_NONE, slope or 0.0)
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    g = torch.Generator().manual_seed(cout + k * 7 + h)
    wk = torch.randn(k, k, 3, cout, generator=g) * (2.0 / (k * k * 3)) ** 0.5
    bk = torch.randn(cout, generator=g) * 0.1
    ps.set_weights({"c/kernel": wk.numpy(), "c/bias": bk.numpy()})
    x = torch.randint(0, 256, (n, 3, h, w), generator=g).float() / 127.5 - 1
    z = K.conv2d(_bf16_round(x), _bf16_round(wk), bk.double(), stride, padding)
    yr = K.leaky_relu(z, slope) if slope else z
    y, ctx = layer.forward(x.to(rt.device).contiguous())
    assert tuple(y.shape) == (n, yr.shape[2], yr.shape[3], cout)
    e_y = rel_err(_to_nchw_f32(rt, y), yr)
    # backward: dz given in bf16 NHWC
    dz = _bf16_round(torch.randn(*yr.shape, generator=g)).float()
    # the weight gradient multiplies dz by the bf16 copy of the frames (the forward's operand: vcg_conv3ch_bf16_wgrad); for odd widths the
    # fp32 kernel on the fp32 frames serves
    bf16_wgrad = rt.lib.vcg_conv3ch_bf16_wgrad_workspace_bytes(ctypes.byref(layer.desc(n, h, w))) > 0
    xg = (_bf16_round(x) if bf16_wgrad else x).double().requires_grad_(True)
    wg = wk.double().requires_grad_(True)
    bg = bk.double().requires_grad_(True)
    (K.conv2d(xg, wg, bg, stride, padding) * dz.double()).sum().backward()
    xr = x.double().requires_grad_(True)          # the data gradient multiplies by the bf16 copy of the kernel and is stored in bf16 on its way
    (K.conv2d(xr, _bf16_round(wk), bk.double(), stride, padding) * dz.double()).sum().backward()
    dx = layer.backward(ctx, _to_nhwc_bf16(rt, dz.to(rt.device)), True, True, 0)
    e_dx, e_dw, e_db = rel_err(dx, xr.grad), rel_err(ps.grad("c/kernel"), wg.grad), rel_err(ps.grad("c/bias"), bg.grad)
    report("first conv bf16 3->%d k%d s%d pad=%s n=%d %dx%d: fwd=%.2e dgrad=%.2e wgrad=%.2e dbias=%.2e" % (cout, k, stride, padding, n, h, w, e_y, e_dx, e_dw, e_db))
    assert e_y < TOL_BF16 and e_dx < TOL_BF16
    assert e_dw < 1e-5 and e_db < 1e-5


# ---------------------------------------------------------------------------------------------------------------
# bf16 discriminators and the all-bf16 train step (BASELINE.json configs C3 / C4)
# ---------------------------------------------------------------------------------------------------------------
def _l2(a, b, floor=0.0):
    return float((a - b).norm() / (b.norm() + floor))


@pytest.mark.parametrize("kind", ["patch", "simple", "thin"])
def test_bf16_discriminator_forward_and_gradients(rt, kind):
    """make_discriminator_*(..., dtype='bf16'): training-mode forward, input gradient and every parameter gradient against the
    fp64 oracle evaluated with the same storage roundings; yardstick per tensor = that tensor's own distance between the fp32
    and fp64 runs of the same emulation (bound 2.5x, floor 1e-2 or half the median such distance over the network's tensors;
    numerically-zero gradients excluded and reported)."""
    from oracle import models as M
    from upscaler import model as PM, _engine as E
    # the Dense-head critics halve the map eight times: at 64x64 their last three blocks are BatchNormalizations over n values on 1x1 maps,
    # which makes ANY fp32 evaluation chaotic (the fp32 oracle itself was 25-60 % off its fp64 run at n = 4): 16 frames of 128x128 for them
    n, hw = (4, 64) if kind == "patch" else (16, 128)
    if kind == "patch":
        D = PM.make_discriminator_patchgan_70((hw, hw, 3), dtype="bf16")
        dfw = lambda w, x, b: M.discriminator_patchgan_70_forward(w, x, True, bf16=b)[0]
    else:
        D = (PM.make_discriminator_simple_512 if kind == "simple" else PM.make_discriminator_thin_512)((hw, hw, 3), dtype="bf16")
        dfw = lambda w, x, b: M.discriminator_512_forward(w, x, True, bf16=b)[0]
    wd = _randomize_bn(D, 7)
    x = (np.random.RandomState(3).randint(0, 256, (n, hw, hw, 3)) / 127.5 - 1).astype(np.float32)
    coef = np.random.RandomState(4).randn(*([n] + list(D.output_shape[1:]))).astype(np.float32)

    def oracle(bf, dt):
        leaf = M.to_torch(wd, dt, requires_grad=True)
        xi = torch.tensor(x, dtype=dt, requires_grad=True)
        y = dfw(leaf, xi, bf)
        loss = (y * torch.tensor(coef, dtype=dt).view(*y.shape)).sum()
        names = [k for k, v in leaf.items() if v.requires_grad]
        gs = torch.autograd.grad(loss, [leaf[k] for k in names] + [xi])
        return y.detach().double(), dict(zip(names, [g.double() for g in gs[:-1]])), gs[-1].double()
    yr, gref, dxr = oracle(True, torch.float64)
    y32, g32, dx32 = oracle(True, torch.float32)
    yp, _, _ = oracle(False, torch.float64)
    xd = E.to_device_nchw(rt, x)
    y, tape = D.forward(xd, True, True)
    dy = E.to_device_nchw(rt, coef.reshape(n, *yr.shape[1:])) if y.dim() == 4 else torch.tensor(coef, device=rt.device).view(*y.shape).contiguous()
    dx = D.backward(tape, dy, True, True, 0)
    yd = (E.to_nhwc(rt, y) if y.dim() == 4 else y).cpu().double()
    e_y, e32_y = _l2(yd, yr), _l2(y32, yr)
    e_dx, e32_dx = _l2(E.to_nhwc(rt, dx).cpu().double(), dxr), _l2(dx32, dxr)
    gmax = max(float(g.abs().max()) for g in gref.values())
    worst = 0.0
    floors = {k: 1e-4 * gmax * b.numel() ** 0.5 for k, b in gref.items()}
    # the error LEVEL of this network in any fp32 evaluation (BatchNormalization over a handful of samples on the 1x1 maps of the last
    # blocks makes the Dense-head critics chaotic, fact (i) of DESIGN.md section 5): a tensor whose own fp32-vs-fp64 distance happens to be
    # ~0 (a gradient that is a plain sum, like the last beta) is still computed from tensors that carry that level
    e32_all = sorted(_l2(g32[k], b, floors[k]) for k, b in gref.items() if float(b.norm()) >= floors[k])
    level = e32_all[len(e32_all) // 2]
    for k, b in gref.items():
        a = D.ps.grad(k).cpu().double()
        floor = floors[k]
        real = float(b.norm()) >= floor
        e, e32 = _l2(a, b, floor), _l2(g32[k], b, floor)
        report("    bf16 D[%s] %-44s |g|2=%.2e rel L2 err=%.2e (oracle fp32-vs-fp64, same storage: %.2e)%s"
               % (kind, k, float(b.norm()), e, e32, "" if real else "   [zero gradient: excluded]"))
        if real:
            worst = max(worst, e)
            assert e < max(1e-2, 2.5 * e32, 0.5 * level), (k, e, e32, level)
    report("bf16 discriminator %s: output err=%.2e (yardstick %.2e; vs un-rounded oracle %.2e)  input-gradient err=%.2e (yardstick %.2e)  worst parameter gradient=%.2e"
           % (kind, e_y, e32_y, _l2(yd, yp), e_dx, e32_dx, worst))
    assert e_y < max(2e-3, 2.5 * e32_y) and e_dx < max(1e-2, 2.5 * e32_dx)


def test_all_bf16_train_step_matches_emulating_oracle(rt):
    """C3's arithmetic at C1's frame size: generator 'bf16+tail' and PatchGAN dtype='bf16' through two loop-body iterations
    (train_gan3.py:346-354) against the fp64 oracle with the same storage roundings in G and D.  Losses of the first iteration
    to max(2e-3, 3x the oracle's own fp32-vs-fp64 distance); second iteration (after one primed Adam update each) to
    max(1e-2, 3x)."""
    from oracle import models as M, train as T
    from upscaler import model as PM, _lib as L
    res, bs, h = 2, 4, 64
    G = PM.make_upscaler_orig((2 * h, 2 * h, 3), kernel_size=3, upscale_factor=2, res_block_num=res, seed=7, trunk_dtype="bf16+tail")
    D = PM.make_discriminator_patchgan_70((2 * h, 2 * h, 3), seed=11, dtype="bf16")
    gw, dw = _randomize_bn(G, 5), _randomize_bn(D, 6)
    gf = lambda w, x, t: M.upscaler_orig_forward(w, x, t, res, 2, trunk_bf16=True, tail_bf16=True)
    df = lambda w, x, t: M.discriminator_patchgan_70_forward(w, x, t, bf16=True)

    def mk(dt):
        return T.GanOracle(gf, M.to_torch(gw, dt), df, M.to_torch(dw, dt), wiring="gan2", content="mse", losses="wass",
                           discriminator_loss_weight=1e-2, adam_v0=1.0)
    orc, orc32 = mk(torch.float64), mk(torch.float32)
    _, _, gan_train = PM.make_and_compile_gan2(G, D, (h, h, 3), (2 * h, 2 * h, 3), "mse", 1.0, lambda: PM.WassersteinLosses(), 1e-2,
                                               optimizer=PM.Adam())
    tr = gan_train.trainer
    for s in (tr.g_slots, tr.d_slots):
        L.check(rt.lib.vcg_fill(s.v.data_ptr(), s.v.numel(), 1.0, rt.stream), "vcg_fill")
    rng = np.random.RandomState(8)
    for it in range(2):
        lr = (rng.randint(0, 256, (bs, h, h, 3)) / 127.5 - 1).astype(np.float32)
        hr = (rng.randint(0, 256, (bs, 2 * h, 2 * h, 3)) / 127.5 - 1).astype(np.float32)
        got = gan_train.train_step(lr, hr)
        ref = orc.train_step(torch.tensor(lr, dtype=torch.float64), torch.tensor(hr, dtype=torch.float64))
        r32 = orc32.train_step(torch.tensor(lr), torch.tensor(hr))
        scale = max(abs(v) for v in ref) + 1e-6
        for name, a, b, c in zip(("disc", "gan", "content", "adv"), got, ref, r32):
            err, e32 = abs(a - b) / scale, abs(c - b) / scale
            report("all-bf16 train step it=%d loss_%s got=%.6g ref=%.6g err=%.1e (oracle fp32-vs-fp64, same storage: %.1e)" % (it, name, a, b, err, e32))
            assert err < max(2e-3 if it == 0 else 1e-2, 3 * e32), (it, name, a, b, c)


def test_all_bf16_train_step_at_c4_frame_size(rt):
    """BASELINE.json config C4 in its stated arithmetic: 540x960 -> 1080x1920 frames, generator 'bf16+tail' + PatchGAN 'bf16',
    one whole train step at batch 1 (finite losses, close to the fp32 product's on the same frames and weights), and the
    inference output of a 32-row band against the oracle (rows far from the band edges, as in
    test_model_gpu.py::test_c4_frame_size_train_step_runs)."""
    from oracle import models as M
    from upscaler import model as PM
    h, w = 540, 960

    def build(bf):
        G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=2, seed=7, trunk_dtype="bf16+tail" if bf else "fp32")
        D = PM.make_discriminator_patchgan_70((2 * h, 2 * w, 3), seed=11, dtype="bf16" if bf else "fp32")
        _, _, gan = PM.make_and_compile_gan2(G, D, (h, w, 3), (2 * h, 2 * w, 3), "mse", 1.0, lambda: PM.WassersteinLosses(), 1e-5, optimizer=PM.Adam())
        return G, D, gan
    rng = np.random.RandomState(9)
    lr = (rng.randint(0, 256, (1, h, w, 3)) / 127.5 - 1).astype(np.float32)
    hr = (rng.randint(0, 256, (1, 2 * h, 2 * w, 3)) / 127.5 - 1).astype(np.float32)
    Gb, Db, ganb = build(True)
    lb = ganb.train_step(lr, hr)
    predb = Gb.predict(lr)                       # after the step: the same weights the oracle band run gets
    gw = M.to_torch(Gb.get_weights_dict(), torch.float64)
    band = lr[:, 262:294]
    with torch.no_grad():
        ref, _ = M.upscaler_orig_forward(gw, torch.tensor(band, dtype=torch.float64), False, 2, 2)
    inner = slice(2 * 14, 2 * 18)
    e = rel_err(predb[:, 2 * 262 + inner.start:2 * 262 + inner.stop], ref[:, inner].numpy())
    Gf, Df, ganf = build(False)
    lf = ganf.train_step(lr, hr)
    report("C4 frame size, all-bf16 step: losses %s (fp32 product %s)  band parity of predict vs fp64 oracle err=%.2e"
           % (["%.5g" % v for v in lb], ["%.5g" % v for v in lf], e))
    assert all(np.isfinite(v) for v in lb)
    assert e < 1e-2                                # bf16 storage through 2 blocks + tail: observed 3.1e-3
    for a, b in zip(lb, lf):
        assert abs(a - b) < 2e-2 * (max(abs(v) for v in lf) + 1e-6), (lb, lf)      # observed 6.5e-3 (adversarial term)


# ---------------------------------------------------------------------------------------------------------------
# statistics of the normalisation behind a convolution, out of the convolution's epilogue (no separate pass over its output)
# ---------------------------------------------------------------------------------------------------------------
def _check_partials(rt, y_nhwc, stats, instance, layer_name, gamma=None, beta=None, tol=2e-5):
    """vcg_norm_finalize_partials on the records `stats` against fp64 statistics of the STORED bf16 tensor y"""
    from upscaler import _engine as E, _lib as L
    buf, nrec = stats
    n, h, w, c = y_nhwc.shape
    rows = n if instance else 1
    cnt = h * w if instance else n * h * w
    mean, scale, shift, invstd = (rt.empty(rows * c) for _ in range(4))
    mm, mv = E.filled_like(rt, rt.empty(c), 0.25), E.filled_like(rt, rt.empty(c), 2.0)
    eps = E.IN_EPS if instance else E.BN_EPS
    L.check(rt.lib.vcg_norm_finalize_partials(buf.data_ptr(), nrec, rows, c, float(cnt), gamma.data_ptr() if gamma is not None else None,
                                              beta.data_ptr() if beta is not None else None, eps, mean.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                              invstd.data_ptr(), None if instance else mm.data_ptr(), None if instance else mv.data_ptr(), 0.99,
                                              0 if instance else cnt, rt.stream), "vcg_norm_finalize_partials")
    yd = y_nhwc.cpu().double()
    dims = (1, 2) if instance else (0, 1, 2)
    m_ref = yd.mean(dims).reshape(-1)
    v_ref = yd.var(dims, unbiased=False).reshape(-1)
    is_ref = 1.0 / torch.sqrt(v_ref + eps)
    sd = float(v_ref.sqrt().max())
    e_m = float((mean.cpu().double() - m_ref).abs().max()) / sd
    e_is = rel_err(invstd, is_ref)
    ga = gamma.cpu().double() if gamma is not None else torch.ones(c, dtype=torch.float64)
    be = beta.cpu().double() if beta is not None else torch.zeros(c, dtype=torch.float64)
    sc_ref = (ga.repeat(rows) * is_ref)
    e_sc = rel_err(scale, sc_ref)
    e_sh = float((shift.cpu().double() - (be.repeat(rows) - m_ref * sc_ref)).abs().max()) / float(1.0 + (m_ref * sc_ref).abs().max())
    report("stats epilogue %s %s n=%d %dx%dx%d: mean=%.1e invstd=%.1e scale=%.1e shift=%.1e (records per group %d)"
           % (layer_name, "instance" if instance else "batch", n, h, w, c, e_m, e_is, e_sc, e_sh, nrec))
    assert e_m < tol and e_is < tol and e_sc < tol and e_sh < tol, (e_m, e_is, e_sc, e_sh)
    if not instance:            # moving averages: momentum 0.99, Bessel-corrected variance (the 4-D Keras path)
        mm_ref = 0.25 * 0.99 + m_ref * 0.01
        mv_ref = 2.0 * 0.99 + v_ref * cnt / (cnt - 1) * 0.01
        assert rel_err(mm, mm_ref) < 1e-6 and rel_err(mv, mv_ref) < 1e-6


@pytest.mark.parametrize("n,h,w,instance", [(2, 16, 32, False), (1, 13, 45, False), (3, 40, 72, True), (2, 5, 7, True), (16, 81, 97, False),
                                             (10, 100, 100, True), (3, 200, 300, False), (8, 256, 256, False)])
def test_conv3x3_c64_bf16_stats_epilogue(rt, n, h, w, instance):
    """Conv3x3Bf16.forward_stats: the output is bit-identical to forward's, and the records finalize to the mean / variance of the
    stored tensor (ragged tiles, tiles smaller than the image, several tiles per workgroup, per-image statistics)"""
    from upscaler import _engine as E
    layer = E.Conv3x3Bf16("c")
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    g = torch.Generator().manual_seed(n * 100 + h + w)
    ps.set_weights({"c/kernel": (torch.randn(3, 3, 64, 64, generator=g) * 0.06).numpy(), "c/bias": (torch.randn(64, generator=g) * 0.7).numpy()})
    x = _to_nhwc_bf16(rt, (torch.randn(n, 64, h, w, generator=g) + 0.3).to(rt.device))
    y0, _ = layer.forward(x)
    y1, _, st = layer.forward_stats(x, instance)
    assert st is not None
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    gamma = (torch.rand(64, generator=g) + 0.5).to(rt.device)
    beta = torch.randn(64, generator=g).to(rt.device)
    _check_partials(rt, y1, st, instance, "conv3x3 v2", None if instance else gamma, None if instance else beta)
    # deterministic
    y2, _, st2 = layer.forward_stats(x, instance)
    assert torch.equal(st[0], st2[0])


GSTAT_CASES = [
    # cin, cout, k, stride, padding, n, h, w, instance
    (64, 128, 4, 2, 1, 4, 128, 96, True),        # PatchGAN block 2 (parity planes)
    (128, 256, 4, 2, 1, 4, 128, 128, True),      # block 3
    (256, 512, 4, 1, 1, 2, 70, 50, True),        # block 4, ragged tiles
    (256, 512, 4, 1, 1, 2, 70, 50, False),       # the same with batch statistics (norm='batch')
    (64, 128, 3, 2, "same", 4, 128, 128, False),  # simple_512 block 2
    (128, 128, 3, 1, "same", 5, 67, 45, False),  # odd sizes, two input chunks
    (64, 192, 3, 1, "same", 4, 64, 64, False),   # a ragged 128-channel group
]


@pytest.mark.parametrize("cin,cout,k,stride,padding,n,h,w,instance", GSTAT_CASES)
def test_generic_conv_bf16_stats_epilogue(rt, cin, cout, k, stride, padding, n, h, w, instance):
    from upscaler import _engine as E
    layer = E.Conv2DBf16("c", cin, cout, k, stride, padding)
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    g = torch.Generator().manual_seed(cin + cout + k + h)
    ps.set_weights({"c/kernel": (torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5).numpy(),
                    "c/bias": (torch.randn(cout, generator=g) * 0.5).numpy()})
    x = _to_nhwc_bf16(rt, torch.randn(n, cin, h, w, generator=g).to(rt.device))
    y0, _ = layer.forward(x)
    y1, _, st = layer.forward_stats(x, instance)
    assert st is not None, "the LDS-tiled kernel serves this shape"
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    _check_partials(rt, y1, st, instance, "gconv_lds %d->%d k%d s%d" % (cin, cout, k, stride))


def test_stats_epilogue_falls_back_when_unsupported(rt):
    """a shape the tiled kernels do not serve (too few tiles) reports no records; the layer then returns stats=None"""
    from upscaler import _engine as E
    layer = E.Conv2DBf16("c", 512, 512, 3, 2, "same")
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    x = _to_nhwc_bf16(rt, torch.randn(2, 512, 4, 4).to(rt.device))
    y, _, st = layer.forward_stats(x, False)
    assert st is None and tuple(y.shape) == (2, 2, 2, 512)
