"""Kernel micro-benchmarks at the C2 shapes (batch 8, 256x256 -> 512x512): times each MFMA kernel family with
HIP events and prints achieved TFLOP/s against the fp32 MFMA peak.  python scripts/kbench.py [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))

import numpy as np
import torch

from upscaler import _engine as E
from upscaler import _lib as L

PEAK = 157.3


def standalone(rt, layer):
    ps = E.ParamStore()
    layer.declare(ps)
    ps.materialize(rt)
    layer.bind(rt, ps)
    ps.set_weights(layer.init_weights(np.random.RandomState(0)))
    return ps


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    rt = E.Runtime.get()
    B = 8
    cases = [
        # name, layer, input shape
        ("trunk 3x3 64->64 @256", E.Conv2D("c", 64, 64, 3), (B, 64, 256, 256)),
        ("first 9x9 3->64 @256", E.Conv2D("c", 3, 64, 9), (B, 3, 256, 256)),
        ("final 9x9 256->3 @512", E.Conv2D("c", 256, 3, 9, act=L.ACT_TANH), (B, 256, 512, 512)),
        ("convT 3x3 64->256 @256", E.ConvT2D("c", 64, 256, 3, L.ACT_LRELU, 0.2), (B, 64, 256, 256)),
        ("pg1 4x4s2 3->64 @512", E.Conv2D("c", 3, 64, 4, 2, 1, L.ACT_LRELU, 0.2), (B, 3, 512, 512)),
        ("pg2 4x4s2 64->128 @256", E.Conv2D("c", 64, 128, 4, 2, 1), (B, 64, 256, 256)),
        ("pg3 4x4s2 128->256 @128", E.Conv2D("c", 128, 256, 4, 2, 1), (B, 128, 128, 128)),
        ("pg4 4x4s1 256->512 @64", E.Conv2D("c", 256, 512, 4, 1, 1), (B, 256, 64, 64)),
        ("pg5 4x4s1 512->1 @63", E.Conv2D("c", 512, 1, 4, 1, 1), (B, 512, 63, 63)),
        ("trunk 5x5 64->64 @256", E.Conv2D("c", 64, 64, 5), (B, 64, 256, 256)),
    ]
    print("%-28s %10s %10s %10s   (ms, TFLOP/s, %% of %.1f)" % ("case", "fwd", "dgrad", "wgrad", PEAK))
    for name, layer, shp in cases:
        if flt and flt not in name:
            continue
        standalone(rt, layer)
        x = torch.randn(*shp, device=rt.device)
        y, ctx = layer.forward(x)
        dy = torch.randn_like(y)
        k = layer.k
        macs = (y.numel() // y.shape[1]) * layer.cin * layer.cout * k * k if not isinstance(layer, E.ConvT2D) else \
            (x.numel() // x.shape[1]) * layer.cin * layer.cout * k * k
        flop = 2.0 * macs
        # isolate the three kernels through the layer API
        act_saved = layer.act
        t_f = timeit(lambda: layer.forward(x))
        layer.act = L.ACT_NONE      # keep the activation backward out of the dgrad / wgrad timings
        ctx2 = (ctx[0], None, ctx[2])
        t_d = timeit(lambda: layer.backward(ctx2, dy, True, False)) if layer.cin > 3 or True else 0
        t_w = timeit(lambda: layer.backward(ctx2, dy, False, True))
        layer.act = act_saved
        f = lambda t: "%6.3f/%5.1f/%2.0f%%" % (t, flop / t / 1e9, 100 * flop / t / 1e9 / PEAK)
        print("%-28s %s %s %s" % (name, f(t_f), f(t_d), f(t_w)), flush=True)
    # HBM-bound passes
    if not flt or "norm" in flt:
        layer = E.NormAct("bn", 64, "batch", L.ACT_PRELU, prelu_name="pr")
        standalone(rt, layer)
        x = torch.randn(B, 64, 256, 256, device=rt.device)
        r = torch.randn_like(x)
        y, ctx = layer.forward(x, True, residual=r)
        dy = torch.randn_like(x)
        gb = x.numel() * 4 / 1e9
        t_f = timeit(lambda: layer.forward(x, True, residual=r))
        t_b = timeit(lambda: layer.backward(ctx, dy))
        print("BN+PReLU+res 64ch @256  fwd %.3f ms (%.0f GB/s alg, 4 passes)  bwd %.3f ms (%.0f GB/s alg, 5 passes)"
              % (t_f, 4 * gb / t_f * 1e3, t_b, 5 * gb / t_b * 1e3))


if __name__ == "__main__":
    main()
