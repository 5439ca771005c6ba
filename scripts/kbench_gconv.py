"""Micro-benchmark of the generic bf16 NHWC convolution (forward / data gradient / weight gradient) at the PatchGAN and
simple_512 shapes of config C2/C3 (batch 8, 512x512 frames).  python scripts/kbench_gconv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))

import numpy as np
import torch

from upscaler import _engine as E

PEAK = 2500.0


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
    rt = E.Runtime.get()
    B = 8
    cases = [("pg2 4x4s2 64->128 @256", 64, 128, 4, 2, 1, 256), ("pg3 4x4s2 128->256 @128", 128, 256, 4, 2, 1, 128),
             ("pg4 4x4s1 256->512 @64", 256, 512, 4, 1, 1, 64), ("s2 3x3s2 64->128 @512", 64, 128, 3, 2, "same", 512),
             ("s3 3x3s2 128->256 @256", 128, 256, 3, 2, "same", 256), ("s4 3x3s2 256->512 @128", 256, 512, 3, 2, "same", 128),
             ("s5 3x3s2 512->512 @64", 512, 512, 3, 2, "same", 64)]
    if len(sys.argv) > 1 and sys.argv[1] == "c4":       # the PatchGAN layers at config C4's frames (1080x1920, batch 4): not L2-resident
        B = 4
        cases = [("pg2 4x4s2 64->128 @540x960", 64, 128, 4, 2, 1, (540, 960)), ("pg3 4x4s2 128->256 @270x480", 128, 256, 4, 2, 1, (270, 480)),
                 ("pg4 4x4s1 256->512 @135x240", 256, 512, 4, 1, 1, (135, 240))]
    print("%-26s %18s %18s %18s   (ms / TFLOP/s / %% of %.0f)" % ("case", "fwd", "dgrad", "wgrad", PEAK))
    for name, cin, cout, k, s, pad, hw in cases:
        hw, ww = hw if isinstance(hw, tuple) else (hw, hw)
        layer = E.Conv2DBf16("c", cin, cout, k, s, pad)
        ps = E.ParamStore()
        layer.declare(ps)
        ps.materialize(rt)
        layer.bind(rt, ps)
        ps.set_weights(layer.init_weights(np.random.RandomState(0)))
        x = torch.randn(B, hw, ww, cin, device=rt.device).to(torch.bfloat16)
        y, ctx = layer.forward(x)
        dy = torch.randn_like(y)
        flop = 2.0 * (y.numel() // cout) * cin * cout * k * k
        t_f = timeit(lambda: layer.forward(x))
        t_d = timeit(lambda: layer.backward(ctx, dy, True, False))
        t_w = timeit(lambda: layer.backward(ctx, dy, False, True))
        f = lambda t: "%6.3f/%6.1f/%4.1f%%" % (t, flop / t / 1e9, 100 * flop / t / 1e9 / PEAK)
        print("%-26s %s %s %s" % (name, f(t_f), f(t_d), f(t_w)), flush=True)


if __name__ == "__main__":
    main()
