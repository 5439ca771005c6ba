"""RCCL smoke of the data-parallel step on ONE GPU: a 1-rank "nccl" process group drives exactly the code path the
N-GPU bench takes (bucket all-reduces between the four hipGraphs, loss scalars reduced at read-out); with one rank
the result must equal the single-process step bit for bit.  python scripts/dp_nccl_smoke.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist


def run(group, graph):
    from upscaler import _engine as E
    from upscaler import model as PM
    G = PM.make_upscaler_orig((64, 64, 3), kernel_size=3, upscale_factor=2, res_block_num=2, seed=7)
    D = PM.make_discriminator_patchgan_70((64, 64, 3), seed=11)
    _, _, gan = PM.make_and_compile_gan2(G, D, (32, 32, 3), (64, 64, 3), "mse", 1.0, lambda: PM.WassersteinLosses(), 1e-2,
                                         optimizer=PM.Adam(), process_group=group)
    tr = gan.trainer
    rt = E.Runtime.get()
    rng = np.random.RandomState(5)
    out = []
    steps = [(E.to_device_nchw(rt, rng.randint(0, 256, (4, 32, 32, 3)) / 127.5 - 1),
              E.to_device_nchw(rt, rng.randint(0, 256, (4, 64, 64, 3)) / 127.5 - 1)) for _ in range(3)]
    if graph:
        run.ngraphs = tr.capture_train_step(*steps[0])
        run.ngraphs = run.ngraphs if isinstance(run.ngraphs, list) else [run.ngraphs]
        for a, b in steps[1:]:
            out.append(tr.train_step_graph(a, b))
    else:
        for a, b in steps:
            out.append(tr.train_step(a, b))
    torch.cuda.synchronize()
    return out[-1], G.ps.params.clone(), D.ps.params.clone()


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    group = dist.group.WORLD
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    assert float(t.sum()) == 4.0
    ref = run(None, False)
    g1 = run(None, True)
    print("single process, one hipGraph: losses %s  identical to eager: %s" % (["%.6g" % v for v in g1[0]],
          torch.equal(g1[1], ref[1]) and torch.equal(g1[2], ref[2])), flush=True)
    for graph in (False, True):
        got = run(group, graph)
        same = torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        print("nccl world=1 %s: losses %s  weights identical to the single-process step: %s"
              % ("%d hipGraphs + eager all-reduces" % len(getattr(run, "ngraphs", [0] * 4)) if graph else "eager", ["%.6g" % v for v in got[0]], same), flush=True)
        assert same and got[0] == ref[0]
    dist.barrier()
    dist.destroy_process_group()
    print("ok")


if __name__ == "__main__":
    main()
