import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd")); sys.path.insert(0, ROOT)
from upscaler import _engine as E, _lib as L
from oracle import keras_ops as K
rt = E.Runtime.get()
for cin, k, pad, n, h, w in ((8, 4, 1, 1, 6, 6), (512, 4, 1, 1, 6, 21)):
    layer = E.ConvCout1Bf16("c", cin, 1, k, 1, pad)
    ps = E.ParamStore(); layer.declare(ps); ps.materialize(rt); layer.bind(rt, ps)
    g = torch.Generator().manual_seed(1)
    wk = torch.randn(k, k, cin, 1, generator=g); bk = torch.zeros(1)
    ps.set_weights({"c/kernel": wk.numpy(), "c/bias": bk.numpy()})
    x = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
    yr = K.conv2d(x.double(), wk.double(), bk.double(), 1, pad)
    xd = x.to(rt.device).permute(0, 2, 3, 1).contiguous().bfloat16()
    y, ctx = layer.forward(xd)
    torch.cuda.synchronize()
    print("case", cin, k, h, w)
    print((y.cpu()[0, 0] - yr[0, 0].float()).abs().numpy().round(3))
