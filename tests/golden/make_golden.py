"""Generates tests/golden/*.npz from the build's own CPU oracle in fp64 (the reference itself -- Keras
2.2.x on TF 1.14 -- cannot be imported here: SURVEY.md section 8c; "parity unpinned").  Fixtures are data only:
seeded inputs are regenerated from the seeds, the expected outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import models as M  # noqa: E402
from oracle import train as T  # noqa: E402

CASES = {
    # BASELINE.json configs[0]: 1 frame 64x64 -> 128x128, 6 res blocks, reference D simple_512, CLI loss weights
    "c1": dict(batch=1, res=6, k=3, disc="simple", losses="wass", dw=1e-5, steps=2),
    # batch 4 / relativistic loss / thin D: exercises BN statistics and the non-linear loss.  adam_v0 primes
    # Adam's second-moment slots (see oracle/train.py): post-update quantities are then well conditioned
    "b4_rel": dict(batch=4, res=2, k=3, disc="thin", losses="rel", dw=1e-2, steps=2, adam_v0=1.0),
    # the north_star PatchGAN extension
    "b2_patch": dict(batch=2, res=2, k=3, disc="patch", losses="wass", dw=1e-2, steps=2, adam_v0=1.0),
}


def frames(seed, n, h, w):
    return np.random.RandomState(seed).randint(0, 256, (n, h, w, 3)) / 127.5 - 1


def build(case):
    c = CASES[case]
    gw = M.init_upscaler_orig((128, 128, 3), c["k"], 64, 2, c["res"], seed=7)
    if c["disc"] == "patch":
        dw = M.init_discriminator_patchgan_70((128, 128, 3), seed=11)
        df = lambda w, x, t: M.discriminator_patchgan_70_forward(w, x, t)
    else:
        dw = M.init_discriminator_512((128, 128, 3), c["disc"], seed=11)
        df = lambda w, x, t: M.discriminator_512_forward(w, x, t)
    gf = lambda w, x, t: M.upscaler_orig_forward(w, x, t, c["res"], 2)
    return c, gw, dw, gf, df


def run(case, dtype=torch.float64):
    c, gw, dw, gf, df = build(case)
    orc = T.GanOracle(gf, M.to_torch(gw, dtype), df, M.to_torch(dw, dtype), wiring="gan2", content="mse",
                      content_loss_weight=1.0, losses=c["losses"], loss_activation="log-sigm",
                      discriminator_loss_weight=c["dw"], adam_v0=c.get("adam_v0", 0.0))
    out = {}
    lr0 = torch.tensor(frames(100, c["batch"], 64, 64), dtype=dtype)
    out["predict0"] = orc.predict(lr0).numpy().astype(np.float32)
    losses = []
    for it in range(c["steps"]):
        lr = torch.tensor(frames(100 + it, c["batch"], 64, 64), dtype=dtype)
        hr = torch.tensor(frames(200 + it, c["batch"], 128, 128), dtype=dtype)
        losses.append(orc.train_step(lr, hr))
    out["losses"] = np.asarray(losses, np.float64)
    out["predict_after"] = orc.predict(lr0).numpy().astype(np.float32)
    for tag, w in (("G", orc.g_w), ("D", orc.d_w)):
        names = list(w.keys())
        out[tag + "_sum"] = np.asarray([float(w[k].sum()) for k in names], np.float64)
        out[tag + "_abs"] = np.asarray([float(w[k].abs().sum()) for k in names], np.float64)
    return out


if __name__ == "__main__":
    for case in CASES:
        res = run(case)
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, **res)
        print(case, os.path.getsize(path) // 1024, "KiB", res["losses"].tolist())
