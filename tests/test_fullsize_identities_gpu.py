"""Size-independent properties at BASELINE.json's FULL sizes (config C2: batch 8, 256x256 -> 512x512), where the CPU
oracle would take minutes: for a linear layer y = A(w) x the three kernels must satisfy

        <dy, fwd(x; w)>  =  <x, dgrad(dy; w)>  =  <w, wgrad(x, dy)>

(adjoint identities; bias off).  Forward, data-gradient and weight-gradient are three different kernels (different
tilings, the row-chain / flat-K / sub-pixel variants per shape), so the identities check them against each other at
the sizes the benchmark runs; the dot products are taken in fp64.  Also: batch-norm output statistics and the
hipGraph step's determinism at full size."""
import numpy as np
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _layer(rt, layer, seed):
    from upscaler import _engine as E
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    w = layer.init_weights(np.random.RandomState(seed))
    for k in w:
        if k.endswith("/bias"):
            w[k] = np.zeros_like(w[k])
    ps.set_weights(w)
    return ps


FULL = [
    # name, kind, cin, cout, k, stride, padding, input n,h,w
    ("trunk 3x3 64->64 @256", "conv", 64, 64, 3, 1, "same", (8, 256, 256)),
    ("first 9x9 3->64 @256", "conv", 3, 64, 9, 1, "same", (8, 256, 256)),
    ("final 9x9 256->3 @512", "conv", 256, 3, 9, 1, "same", (8, 512, 512)),
    ("upsampling convT 3x3 64->256 @256", "convt", 64, 256, 3, 2, None, (8, 256, 256)),
    ("patchgan 4x4 s2 64->128 @256", "conv", 64, 128, 4, 2, 1, (8, 256, 256)),
    ("patchgan 4x4 s1 256->512 @64", "conv", 256, 512, 4, 1, 1, (8, 64, 64)),
    ("patchgan head 4x4 s1 512->1 @63", "conv", 512, 1, 4, 1, 1, (8, 63, 63)),
]


@pytest.mark.parametrize("name,kind,cin,cout,k,stride,padding,shape", FULL)
def test_adjoint_identities_at_c2_size(rt, name, kind, cin, cout, k, stride, padding, shape):
    from upscaler import _engine as E
    layer = E.Conv2D("c", cin, cout, k, stride, padding) if kind == "conv" else E.ConvT2D("c", cin, cout, k)
    ps = _layer(rt, layer, seed=cin + cout)
    n, h, w = shape
    g = torch.Generator(device=rt.device).manual_seed(7)
    x = torch.randn(n, cin, h, w, generator=g, device=rt.device)
    y, ctx = layer.forward(x)
    dy = torch.randn(*y.shape, generator=g, device=rt.device)
    dx = layer.backward(ctx, dy, True, True, 0)
    torch.cuda.synchronize()
    a, b, c = _dot(dy, y), _dot(x, dx), _dot(ps["c/kernel"], ps.grad("c/kernel"))
    scale = float(dy.double().norm() * y.double().norm())
    e1, e2 = abs(a - b) / scale, abs(a - c) / scale
    db = float((ps.grad("c/bias").double() - dy.double().sum((0, 2, 3))).abs().max() / (dy.double().sum((0, 2, 3)).abs().max() + 1e-30))
    report("full-size adjoint %-34s <dy,y>=%.6e  |<x,dx>-.|/(|dy||y|)=%.1e  |<w,dw>-.|/(|dy||y|)=%.1e  dbias err=%.1e" % (name, a, e1, e2, db))
    assert e1 < 1e-6 and e2 < 1e-6 and db < 1e-4


def test_batchnorm_output_statistics_at_c2_size(rt):
    """training-mode BN at [8,64,256,256]: per-channel output mean 0 and variance var/(var+eps) (gamma 1, beta 0)"""
    from upscaler import _engine as E
    layer = E.NormAct("bn", 64, "batch")
    ps = _layer(rt, layer, 1)
    g = torch.Generator(device=rt.device).manual_seed(3)
    x = torch.randn(8, 64, 256, 256, generator=g, device=rt.device) * 2.5 + 1.25
    y, _ = layer.forward(x, True)
    m = y.double().mean((0, 2, 3))
    v = y.double().var((0, 2, 3), unbiased=False)
    xv = x.double().var((0, 2, 3), unbiased=False)
    report("full-size BN: max |mean|=%.1e  max |var - var/(var+eps)|=%.1e" % (float(m.abs().max()), float((v - xv / (xv + 1e-3)).abs().max())))
    assert float(m.abs().max()) < 1e-5 and float((v - xv / (xv + 1e-3)).abs().max()) < 1e-5
