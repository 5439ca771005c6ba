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
    # BASELINE.json configs[1] at FULL frame size (256x256 -> 512x512, 9 res blocks, 70x70 PatchGAN, CLI loss weights),
    # batch 2: the inference output (stored sub-sampled 8x8 plus its sums) and the losses of the first step.  Only with
    # `python make_golden.py c2` (about two minutes of fp64 CPU time).
    "c2": dict(batch=2, res=9, k=3, disc="patch", losses="wass", dw=1e-5, steps=1, lr=256, sub=8),
    # BASELINE.json configs[2]'s arithmetic (C3: bf16 storage in G and D, fp32 accumulation) at its FULL frame size, batch 2: the oracle
    # marks every tensor the product stores in bf16 (keras_ops.bf16_*: upscaler_orig_forward(trunk_bf16, tail_bf16), PatchGAN bf16=True).
    # `python make_golden.py c3`
    "c3": dict(batch=2, res=9, k=3, disc="patch", losses="wass", dw=1e-5, steps=1, lr=256, sub=8, bf16=True),
}


def frames(seed, n, h, w):
    return np.random.RandomState(seed).randint(0, 256, (n, h, w, 3)) / 127.5 - 1


def build(case):
    c = CASES[case]
    hr = 2 * c.get("lr", 64)
    gw = M.init_upscaler_orig((hr, hr, 3), c["k"], 64, 2, c["res"], seed=7)
    bf = bool(c.get("bf16"))
    if c["disc"] == "patch":
        dw = M.init_discriminator_patchgan_70((hr, hr, 3), seed=11)
        df = lambda w, x, t: M.discriminator_patchgan_70_forward(w, x, t, bf16=bf)
    else:
        dw = M.init_discriminator_512((hr, hr, 3), c["disc"], seed=11)
        df = lambda w, x, t: M.discriminator_512_forward(w, x, t, bf16=bf)
    gf = lambda w, x, t: M.upscaler_orig_forward(w, x, t, c["res"], 2, trunk_bf16=bf, tail_bf16=bf)
    return c, gw, dw, gf, df


def run(case, dtype=torch.float64):
    c, gw, dw, gf, df = build(case)
    orc = T.GanOracle(gf, M.to_torch(gw, dtype), df, M.to_torch(dw, dtype), wiring="gan2", content="mse",
                      content_loss_weight=1.0, losses=c["losses"], loss_activation="log-sigm",
                      discriminator_loss_weight=c["dw"], adam_v0=c.get("adam_v0", 0.0))
    out = {}
    n, sub = c.get("lr", 64), c.get("sub", 1)

    def pack(tag, y):          # full tensor, or (full-size cases) an 8x8 sub-sample plus its sums
        y = y.numpy()
        if sub == 1:
            out[tag] = y.astype(np.float32)
        else:
            out[tag + "_sub"] = y[:, ::sub, ::sub].astype(np.float32)
            out[tag + "_sums"] = np.asarray([y.sum(), np.abs(y).sum(), (y * y).sum()], np.float64)
    lr0 = torch.tensor(frames(100, c["batch"], n, n), dtype=dtype)
    pack("predict0", orc.predict(lr0))
    losses = []
    for it in range(c["steps"]):
        lr = torch.tensor(frames(100 + it, c["batch"], n, n), dtype=dtype)
        hr = torch.tensor(frames(200 + it, c["batch"], 2 * n, 2 * n), dtype=dtype)
        losses.append(orc.train_step(lr, hr))
    out["losses"] = np.asarray(losses, np.float64)
    pack("predict_after", orc.predict(lr0))
    for tag, w in (("G", orc.g_w), ("D", orc.d_w)):
        names = list(w.keys())
        out[tag + "_sum"] = np.asarray([float(w[k].sum()) for k in names], np.float64)
        out[tag + "_abs"] = np.asarray([float(w[k].abs().sum()) for k in names], np.float64)
    return out


def randomize_norm(w, seed):
    """non-trivial BatchNormalization statistics / affine parameters / PReLU slopes, as after training (shared by fixture and test)"""
    rng = np.random.RandomState(seed)
    for k, v in w.items():
        if k.endswith("/gamma"):
            w[k] = rng.uniform(0.7, 1.3, v.shape).astype(np.float32)
        elif k.endswith(("/beta", "/moving_mean", "/bias")):
            w[k] = rng.uniform(-0.2, 0.2, v.shape).astype(np.float32)
        elif k.endswith("/moving_variance"):
            w[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith("/alpha"):
            w[k] = rng.uniform(0.0, 0.3, v.shape).astype(np.float32)
    return w


def run_c5():
    """BASELINE.json configs[4] (inference-only generator, bf16, 256 -> 512, 9 blocks): learning-phase-0 forward of FOUR distinct frames with
    the storage roundings of the product's inference engine (BatchNormalization folded: the convolution outputs are never stored), fp64.
    The GPU test replicates the four frames to the configuration's batch of 32 (frames are independent in inference)."""
    gw = randomize_norm(M.init_upscaler_orig((512, 512, 3), 3, 64, 2, 9, seed=7), 3)
    x = torch.tensor(frames(300, 4, 256, 256), dtype=torch.float64)
    with torch.no_grad():
        y, _ = M.upscaler_orig_forward(M.to_torch(gw, torch.float64), x, False, 9, 2, trunk_bf16=True, tail_bf16=True)
        y32, _ = M.upscaler_orig_forward(M.to_torch(gw, torch.float32), x.float(), False, 9, 2, trunk_bf16=True, tail_bf16=True)
    y = y.numpy()
    return {"predict_sub": y[:, ::8, ::8].astype(np.float32),
            "predict_sums": np.stack([np.asarray([f.sum(), np.abs(f).sum(), (f * f).sum()], np.float64) for f in y]),
            # the emulation's own fp32-vs-fp64 distance (bf16 ties flip and are amplified through 21 stored tensors): max-norm and L2
            "fp32_emulation_err": np.asarray([float(np.max(np.abs(y32.double().numpy() - y)) / np.max(np.abs(y))),
                                              float(np.linalg.norm(y32.double().numpy() - y) / np.linalg.norm(y))], np.float64)}


if __name__ == "__main__":
    for case in (sys.argv[1:] or [c for c in CASES if c not in ("c2", "c3", "c5")]):
        res = run_c5() if case == "c5" else run(case)
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, **res)
        print(case, os.path.getsize(path) // 1024, "KiB", res["losses"].tolist() if "losses" in res else "")
