"""Headline benchmark: up-scaled frames/s of the GAN train step at 256->512 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): generator make_upscaler_orig((512,512,3), kernel_size=3,
upscale_factor=2, res_block_num=9) + 70x70 PatchGAN, fp32, batch 8 frames per GPU (weak scaling),
make_and_compile_gan2 wiring with WassersteinLosses, pixel-MSE content loss, loss weights 1 and 1e-5.
A step is one iteration of the reference loop body (train_gan3.py:346-354): gen_train.predict ->
disc_train.train_on_batch -> gan_train.train_on_batch, on synthetic frames already resident in HBM.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel (3x3 64->64 trunk convolution, forward and its data gradient run
                  the same kernel): algorithmic FLOPs per launch / mean launch duration measured with HIP
                  events on the launch stream inside the timed region, against the fp32 MFMA peak.
  cpu_baseline -- the CPU oracle (a port: the Keras/TF reference cannot run offline) timed on this
                  host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "video-cycle_gan-upscaling_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
METRIC = "upscaled frames/s (train step, G+D fwd+bwd) at 256->512"
# HBM traffic of one launch of the dominant kernel at the C2 shape, measured with rocprofv3 PMC counters in
# separate passes (scripts/pmc_kbench.sh "trunk 3x3"): FETCH_SIZE 141.2 MiB x 2 (gfx950 reports half of a
# coalesced read; calibrated on stats_partial_kernel reading 128 MiB -> 64.1) + WRITE_SIZE 128.0 MiB
TRUNK_CONV_HBM_BYTES = int((2 * 141.2 + 128.0) * 1024 * 1024)


class KernelProf:
    def __init__(self, tags):
        self.tags = set(tags)
        self.events = {}

    def mean_ms(self, tags):
        tot, cnt = 0.0, 0
        for t in tags:
            for e0, e1 in self.events.get(t, []):
                tot += e0.elapsed_time(e1)
                cnt += 1
        return (tot / cnt if cnt else None), cnt


def _host_cores():
    """cores this process may actually use: cgroup CPU quota if set, else the affinity mask (a GPU box
    exposes all host cores but grants a share -- 16 per GPU on this pool)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("VCG_CPU_THREADS", "16"))))


def cpu_baseline(res_blocks, sample_batch, lr_hw):
    """One train step of the CPU oracle (torch CPU fp32, all host cores) on `sample_batch` frames of the
    same models / frame size."""
    import numpy as np
    import torch
    from oracle import models as OM, train as OT
    torch.set_num_threads(_host_cores())
    h = lr_hw
    gw = OM.to_torch(OM.init_upscaler_orig((2 * h, 2 * h, 3), 3, 64, 2, res_blocks, seed=7))
    dw = OM.to_torch(OM.init_discriminator_patchgan_70((2 * h, 2 * h, 3), seed=11))
    orc = OT.GanOracle(lambda w, x, t: OM.upscaler_orig_forward(w, x, t, res_blocks, 2), gw,
                       lambda w, x, t: OM.discriminator_patchgan_70_forward(w, x, t), dw,
                       discriminator_loss_weight=1e-5)
    g = torch.Generator().manual_seed(1234)
    lr = torch.randint(0, 256, (sample_batch, h, h, 3), generator=g).float() / 127.5 - 1
    hr = torch.randint(0, 256, (sample_batch, 2 * h, 2 * h, 3), generator=g).float() / 127.5 - 1
    t0 = time.perf_counter()
    orc.train_step(lr, hr)
    dt = time.perf_counter() - t0
    return {"value": round(sample_batch / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 train step (predict + D step + G step) of the torch-CPU fp32 oracle on %d frame(s) %dx%d->%dx%d, "
                      "same models; %.1f s" % (sample_batch, h, h, 2 * h, 2 * h, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU")
    ap.add_argument("--lr-size", type=int, default=256, help="low-res frame edge (output is 2x)")
    ap.add_argument("--lr-width", type=int, default=0, help="low-res frame width if not square (e.g. --lr-size 540 --lr-width 960: config C4's frames)")
    ap.add_argument("--res-blocks", type=int, default=9)
    ap.add_argument("--disc", default="patchgan", choices=["patchgan", "simple"])
    ap.add_argument("--content", default="mse", choices=["mse", "vgg_mse"],
                    help="content loss: pixel MSE (C2 as SURVEY 8d defines it) or the reference's default VGG_MSE_LOSS form with "
                         "seeded random VGG19 weights (ImageNet weights cannot be fetched offline)")
    ap.add_argument("--trunk-dtype", default="fp32", choices=["fp32", "bf16", "bf16+tail"],
                    help="bf16: the generator's residual trunk trains on bf16 activations (mixed precision; NOT config C2, reported as such)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one hipGraph per step")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from upscaler import _dist
    from upscaler import _engine as E
    from upscaler import data as PD
    from upscaler import model as PM

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # VCG_BENCH_REHEARSAL=1: run the N-rank code path on a box with fewer GPUs than ranks (all ranks on cuda:0, gloo
    # instead of RCCL).  For rehearsing the launcher / capture / reduction logic only -- its numbers mean nothing.
    rehearsal = os.environ.get("VCG_BENCH_REHEARSAL") == "1"
    torch.cuda.set_device(0 if rehearsal else local_rank)
    group = _dist.init_from_env("gloo" if rehearsal else "nccl") if world > 1 else None

    h = args.lr_size
    w = args.lr_width or h
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=args.res_blocks, seed=7,
                              trunk_dtype=args.trunk_dtype)
    D = (PM.make_discriminator_patchgan_70((2 * h, 2 * w, 3), seed=11) if args.disc == "patchgan"
         else PM.make_discriminator_simple_512((2 * h, 2 * w, 3), seed=11))
    if group is not None:                     # identical replicas: broadcast rank 0's weights
        for m in (G, D):
            _dist.broadcast_(m.ps.params, group)
            _dist.broadcast_(m.ps.state, group)
            m.refresh()
    content = "mse" if args.content == "mse" else PM.VGG_MSE_LOSS((2 * h, 2 * w, 3), 0.1, vgg19="random").loss
    gen_train, disc_train, gan_train = PM.make_and_compile_gan2(
        G, D, (h, w, 3), (2 * h, 2 * w, 3), content, 1.0, lambda: PM.WassersteinLosses(), 1e-5, optimizer=PM.Adam(),
        process_group=group)
    trainer = gan_train.trainer
    rt = E.Runtime.get()

    # synthetic frames: uint8 U{0..255} -> v/127.5-1 (data.py:266-270), resident in HBM before timing
    g1 = torch.Generator().manual_seed(1234 + rank)
    g2 = torch.Generator().manual_seed(4321 + rank)
    lr = PD.frames_u8_to_device(torch.randint(0, 256, (args.batch, h, w, 3), generator=g1, dtype=torch.uint8))
    hr = PD.frames_u8_to_device(torch.randint(0, 256, (args.batch, 2 * h, 2 * w, 3), generator=g2, dtype=torch.uint8))

    def sync():
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier(group=group)
            torch.cuda.synchronize()

    # One hipGraph per step (Wasserstein losses: no host read inside the step).  Under DP the step is three graphs
    # cut at the two exchange points, with the two RCCL all-reduces issued eagerly between the replays.
    use_graph = not args.no_graph
    step = trainer.train_step
    if use_graph:
        try:
            trainer.capture_train_step(lr, hr)
        except Exception as e:      # capture unsupported on this stack: say so and run eagerly
            print("graph capture failed (%s: %s); running eagerly" % (type(e).__name__, e), file=sys.stderr)
            use_graph = False
        if group is not None:       # all ranks must take the same path
            ok = torch.tensor([1 if use_graph else 0], dtype=torch.int32, device="cpu" if rehearsal else rt.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            use_graph = bool(ok.item())
        if use_graph:
            step = trainer.train_step_graph
    for _ in range(args.warmup):
        step(lr, hr)
    dom_tags = ("trunk_conv", "trunk_conv_dgrad")
    if not use_graph:
        rt.prof = KernelProf(dom_tags)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step(lr, hr)
    sync()
    dt = time.perf_counter() - t0
    prof, rt.prof = rt.prof, None
    roof_note = "HIP events around every launch of the kernel inside the timed region"
    if use_graph:
        # kernels inside a graph replay cannot be bracketed one by one: time the same kernel launches with HIP
        # events in eager steps of the same workload right after the timed region
        rt.prof = prof = KernelProf(dom_tags)
        for _ in range(3):
            trainer.train_step(lr, hr)
        torch.cuda.synchronize()
        rt.prof = None
        roof_note = ("the timed region replays one hipGraph per step, which cannot be bracketed per kernel; these are HIP-event "
                     "timings of the same launches in 3 eager steps of the same workload run right after it")

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else rt.device)
    if group is not None:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX, group=group)
    dt = float(dt_t.item())
    frames = args.batch * world * args.steps

    # roofline of the dominant kernel
    mean_ms, launches = prof.mean_ms(dom_tags)
    flop_per_launch = 2.0 * (64 * 64 * 9) * (h * w) * args.batch           # 2 * MAC/pixel * pixels * frames
    roof = None
    if mean_ms:
        ach = flop_per_launch / (mean_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_fwd_kernel<3,3,1,8> (64->64 3x3 trunk conv, forward + dgrad)",
                "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": TRUNK_CONV_HBM_BYTES if (h, w, args.batch) == (256, 256, 8) else None,
                "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction, calibrated on a "
                                "128 MiB streaming read; + WRITE_SIZE): profiles/r01_pmc_trunk_conv.txt; algorithmic bytes 268.6e6",
                "launches_timed": launches, "mean_launch_ms": round(mean_ms, 4),
                "flop_per_launch": flop_per_launch, "how": roof_note}

    if rank == 0:
        out = {
            "metric": METRIC if (h, w) == (256, 256) else METRIC.replace("256->512", "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w)), "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.trunk_dtype == "fp32" else "generator %s activations bf16 / f32 elsewhere (mixed; not C2)" % ("trunk" if args.trunk_dtype == "bf16" else "trunk + up-sampling + final conv"), "data": "synthetic",
            "config": {"workload": ("C2" if (h, w) == (256, 256) else "C4 frame size" if (h, w) == (540, 960) else "custom size") + ": make_upscaler_orig((%d,%d,3),k=3,x2,res=%d) + %s, batch %d/GPU, "
                                   "gan2 wiring, Wasserstein + %s, faithful 3-call step incl. predict pass"
                                   % (2 * h, 2 * w, args.res_blocks, "PatchGAN-70" if args.disc == "patchgan" else "simple_512",
                                      args.batch, "pixel-MSE" if args.content == "mse" else "VGG_MSE_LOSS(0.1), random VGG19 weights"),
                       "global_batch": args.batch * world, "frame": "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w),
                       "parallelism": "dp%d" % world, "launch": ("hipGraph replay" if world == 1 else "3 hipGraphs per step around the 2 RCCL all-reduces") if use_graph else "eager"},
            "last_losses": [round(float(v), 6) for v in losses],
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and args.content == "mse" and args.trunk_dtype == "fp32" and w == h:
            out["cpu_baseline"] = cpu_baseline(args.res_blocks, args.cpu_sample_batch, h)
        print(json.dumps(out), flush=True)
    if group is not None:
        dist.barrier(group=group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
