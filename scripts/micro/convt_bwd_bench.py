import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))
import numpy as np, torch
from upscaler import _engine as E, _lib as L
rt = E.Runtime.get()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = E.ConvT3x3Bf16("u", 64, 256, 3, L.ACT_LRELU, 0.2)
ps = E.ParamStore(); up.declare(ps); ps.materialize(rt); up.bind(rt, ps); ps.set_weights(up.init_weights(np.random.RandomState(0)))
x = torch.randn(B, 256, 256, 64, device=rt.device).to(torch.bfloat16)
y, ctx = up.forward(x)
dy = torch.randn_like(y)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("convT bwd dgrad only: %.1f us   wgrad only: %.1f us   (VCG_GCONV_PLANE_INNER=%s)" % (t(lambda: up.backward(ctx, dy, True, False)), t(lambda: up.backward(ctx, dy, False, True)), os.environ.get("VCG_GCONV_PLANE_INNER", "0")))
