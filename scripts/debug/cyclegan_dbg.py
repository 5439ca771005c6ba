import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import generators as OG, models as M
from upscaler import _engine as E, model as PM
import test_generators_gpu as T
rt = E.Runtime.get()
for norm, f, nd, res in (("instance", 1, 2, 3), ("batch", 2, 1, 2), ("batch", 1, 2, 1), ("instance", 1, 1, 0), ("instance", 1, 0, 1)):
    h, w = 24, 40
    okw = dict(filters=64, n_downsample=nd, res_block_num=res, upscale_factor=f, norm=norm)
    G = PM.make_generator_cyclegan((h * f, w * f, 3), seed=3, **okw)
    gw = T._randomise(OG.init_weights(OG.generator_cyclegan, (h, w, 3), 31, **okw), 32)
    G.set_weights_dict(gw)
    rng = np.random.RandomState(33)
    x = (rng.randint(0, 256, (2, h, w, 3)) / 127.5 - 1).astype(np.float32)
    t = (rng.randint(0, 256, (2, h * f, w * f, 3)) / 127.5 - 1).astype(np.float32)
    yd, tape = G.forward(E.to_device_nchw(rt, x), True)
    leaf = M.to_torch(gw, torch.float64, requires_grad=True)
    y = OG.generator_cyclegan(OG.Net(leaf, True), torch.tensor(x, dtype=torch.float64), **okw)
    names = [n for n, v in leaf.items() if v.requires_grad]
    loss = ((y - torch.tensor(t, dtype=torch.float64)) ** 2).mean()
    grads = dict(zip(names, torch.autograd.grad(loss, [leaf[n] for n in names])))
    val, dy = PM._pixel_loss(rt, yd, E.to_device_nchw(rt, t), "mse", 1.0)
    G.backward(tape, dy, 0)
    gmax = max(float(g.abs().max()) for g in grads.values())
    print("==", norm, f, nd, res, "fwd err", float((E.to_nhwc(rt, yd).cpu().double() - y).abs().max()))
    for n in names:
        a, b = G.ps.grad(n).cpu().double().reshape(grads[n].shape), grads[n]
        err = float((a - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
        if err > 1e-4:
            print("   %-40s |g|=%.2e err=%.2e" % (n, float(b.abs().max()), err))
