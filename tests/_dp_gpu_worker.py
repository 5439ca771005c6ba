"""worker of tests/test_dp_gpu.py: one data-parallel rank of the product trainer (gloo ranks sharing cuda:0,
or the single-process run on the whole batch when WORLD_SIZE is 1).  Writes its final weights to argv[1]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))


def main():
    out, mode = sys.argv[1], sys.argv[2]          # mode: eager | graph | eager-bf16 | graph-bf16 | graph-rel | eager-fused | graph-fused
    bf16 = mode.endswith("-bf16")
    rel = mode.endswith("-rel")
    fused = mode.endswith("-fused")
    mode = mode.split("-")[0]
    from upscaler import _dist
    from upscaler import _engine as E
    from upscaler import model as PM
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    group = _dist.init_from_env("gloo")
    G = PM.make_upscaler_orig((64, 64, 3), kernel_size=3, upscale_factor=2, res_block_num=2, norm="instance", seed=7,
                              trunk_dtype="bf16+tail" if bf16 else "fp32")
    D = PM.make_discriminator_patchgan_70((64, 64, 3), seed=11, dtype="bf16" if bf16 else "fp32")
    fac = (lambda: PM.RelativisticLosses(loss_activation="log-sigm")) if rel else (lambda: PM.WassersteinLosses())
    _, _, gan_train = PM.make_and_compile_gan2(G, D, (32, 32, 3), (64, 64, 3), "mse", 1.0, fac,
                                               1e-2, optimizer=PM.Adam(), process_group=group, fused_step=fused)
    tr = gan_train.trainer
    tr.g_slots.v.fill_(1.0)          # well-conditioned Adam (see tests/test_model_gpu.py)
    tr.d_slots.v.fill_(1.0)
    rt = E.Runtime.get()
    w0 = {"G/" + k: v.copy() for k, v in G.get_weights_dict().items()}
    w0.update({"D/" + k: v.copy() for k, v in D.get_weights_dict().items()})
    rng = np.random.RandomState(5)
    steps = []
    for _ in range(3):
        lr = rng.randint(0, 256, (4, 32, 32, 3)) / 127.5 - 1
        hr = rng.randint(0, 256, (4, 64, 64, 3)) / 127.5 - 1
        lo, hi = _dist.shard_batch(4, group)
        steps.append((E.to_device_nchw(rt, lr[lo:hi]), E.to_device_nchw(rt, hr[lo:hi])))
    losses = []
    if mode == "graph":
        tr.capture_train_step(*steps[0])           # one real eager step on steps[0], then records
        for a, b in steps[1:]:
            losses.append(tr.train_step_graph(a, b))
    else:
        for a, b in steps:
            losses.append(tr.train_step(a, b))
    torch.cuda.synchronize()
    if int(os.environ.get("RANK", "0")) == 0:
        w = {"G/" + k: v for k, v in G.get_weights_dict().items()}
        w.update({"D/" + k: v for k, v in D.get_weights_dict().items()})
        w.update({"init/" + k: v for k, v in w0.items()})
        w["losses"] = np.asarray(losses[-2:], dtype=np.float64)
        np.savez(out, **w)
    if group is not None:
        import torch.distributed as dist
        dist.barrier(group=group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
