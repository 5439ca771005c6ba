"""Headline benchmark: up-scaled frames/s of the GAN train step at 256->512 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): generator make_upscaler_orig((512,512,3), kernel_size=3,
upscale_factor=2, res_block_num=9) + 70x70 PatchGAN, fp32, batch 8 frames per GPU (weak scaling),
make_and_compile_gan2 wiring with WassersteinLosses, pixel-MSE content loss, loss weights 1 and 1e-5.
A step is one iteration of the reference loop body (train_gan3.py:346-354): gen_train.predict ->
disc_train.train_on_batch -> gan_train.train_on_batch, on synthetic frames already resident in HBM.

Secondary lines (same JSON format):
    --disc simple             the reference's own make_discriminator_simple_512 instead of the PatchGAN
    --kernel-size 5           the reference's default residual-block kernel (model.py:267)
    --gan-losses rel --disc-activation bi-log      the reference's default loss configuration (train_gan3.py:58,62-63)
    --dtype bf16              C3/C4's arithmetic: bf16 activations in G and D, fp32 accumulation / statistics / master weights
    --config c5               BASELINE.json configs[4]: inference-only generator, batch 32, bf16, one hipGraph replay per batch

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel: algorithmic FLOPs (or bytes) per launch / mean launch duration measured with HIP
                  events on the launch stream, against the peak that bounds it; `traffic` = measured HBM bytes per launch
                  looked up BY KERNEL NAME AND SHAPE in profiles/pmc_traffic.json (written by scripts/pmc_to_json.py from
                  separate rocprofv3 --pmc passes), null when no PMC record matches the workload.
  cpu_baseline -- the CPU oracle (a port: the Keras/TF reference cannot run offline) timed on this
                  host's cores on a bounded sample of the same workload (1 warm-up + 2 timed steps).
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
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same guide: ~2.5 PF dense
PEAK_HBM_GBS = 8000.0               # same guide: HBM3E 8.0 TB/s spec
METRIC = "upscaled frames/s (train step, G+D fwd+bwd) at 256->512"


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


def pmc_traffic(kernel, shape_key):
    """measured HBM bytes per launch of `kernel` at `shape_key` from profiles/pmc_traffic.json, or (None, None)"""
    try:
        recs = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    for r in recs:
        if r["kernel"] == kernel and r["shape"] == shape_key:
            return int(r["fetch_bytes"] + r["write_bytes"]), r
    return None, None


def pmc_step_traffic(shape_key):
    """measured HBM bytes of a WHOLE step at `shape_key` (scripts/pmc_step.sh -> scripts/pmc_step_summary.py --merge), or None"""
    try:
        for st in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("steps", []):
            if st["shape"] == shape_key:
                return st
    except (OSError, ValueError, KeyError):
        pass
    return None


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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, h, w):
    """The CPU oracle (torch CPU fp32, this host's core share) on `--cpu-sample-batch` frames of the same models / frame
    size / losses: 1 untimed warm-up step, then 2 timed train steps."""
    import torch
    from oracle import models as OM, train as OT
    torch.set_num_threads(_host_cores())
    nb, k, res = args.cpu_sample_batch, args.kernel_size, args.res_blocks
    gw = OM.to_torch(OM.init_upscaler_orig((2 * h, 2 * w, 3), k, 64, 2, res, seed=7))
    if args.disc == "patchgan":
        dw = OM.to_torch(OM.init_discriminator_patchgan_70((2 * h, 2 * w, 3), seed=11))
        df = lambda wt, x, t: OM.discriminator_patchgan_70_forward(wt, x, t, activation=args.disc_activation)
    else:
        dw = OM.to_torch(OM.init_discriminator_512((2 * h, 2 * w, 3), "simple", seed=11))
        df = lambda wt, x, t: OM.discriminator_512_forward(wt, x, t, activation=args.disc_activation)
    orc = OT.GanOracle(lambda wt, x, t: OM.upscaler_orig_forward(wt, x, t, res, 2), gw, df, dw, losses=args.gan_losses,
                       loss_activation="log-sigm", discriminator_loss_weight=1e-5)
    g = torch.Generator().manual_seed(1234)
    lr = torch.randint(0, 256, (nb, h, w, 3), generator=g).float() / 127.5 - 1
    hr = torch.randint(0, 256, (nb, 2 * h, 2 * w, 3), generator=g).float() / 127.5 - 1
    orc.train_step(lr, hr)                       # warm-up (allocator, oneDNN primitive caches)
    timed = 2
    t0 = time.perf_counter()
    for _ in range(timed):
        orc.train_step(lr, hr)
    dt = time.perf_counter() - t0
    return {"value": round(nb * timed / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": _cpu_model(),
            "sample": "%d timed train steps (predict + D step + G step) after 1 warm-up step of the torch-CPU fp32 oracle on %d "
                      "frame(s) %dx%d->%dx%d, same models and losses; %.1f s timed" % (timed, nb, h, w, 2 * h, 2 * w, dt)}


def cpu_baseline_c5(args, h, w):
    import torch
    from oracle import models as OM
    torch.set_num_threads(_host_cores())
    gw = OM.to_torch(OM.init_upscaler_orig((2 * h, 2 * w, 3), 3, 64, 2, args.res_blocks, seed=7))
    g = torch.Generator().manual_seed(1234)
    nb = 4
    x = torch.randint(0, 256, (nb, h, w, 3), generator=g).float() / 127.5 - 1
    with torch.no_grad():
        OM.upscaler_orig_forward(gw, x, False, args.res_blocks, 2)
        t0 = time.perf_counter()
        for _ in range(3):
            OM.upscaler_orig_forward(gw, x, False, args.res_blocks, 2)
        dt = time.perf_counter() - t0
    return {"value": round(3 * nb / dt, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": _cpu_model(),
            "sample": "3 timed predict passes of %d frames after 1 warm-up, torch-CPU fp32 oracle, same generator; %.1f s" % (nb, dt)}


def run_c5(args):
    """BASELINE.json configs[4]: generator.predict only, bf16 NHWC activations, BN folded, one hipGraph replay per batch"""
    import torch
    from upscaler import _engine as E
    from upscaler import data as PD
    from upscaler import model as PM

    h, B = args.lr_size, args.batch or 32
    w = args.lr_width or h
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=args.res_blocks, seed=7)
    inf = G.to_inference_bf16()
    g1 = torch.Generator().manual_seed(1234)
    x = PD.frames_u8_to_device(torch.randint(0, 256, (B, h, w, 3), generator=g1, dtype=torch.uint8))
    step = inf.forward if args.no_graph else inf.replay       # --no-graph: eager launches (the PMC passes of scripts/pmc_step.sh)
    if not args.no_graph:
        inf.capture(B, h, w)
    for _ in range(args.warmup):
        step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # dominant kernel: the 64->64 3x3 trunk convolution (2*res+1 launches per batch); HIP events around eager launches
    evs = []
    orig = inf._conv

    def timed(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(*a)
        e1.record()
        evs.append((e0, e1, a[7] is not None))
    inf._conv = timed
    for _ in range(3):
        inf.forward(x)
    torch.cuda.synchronize()
    inf._conv = orig
    tensor_bytes = B * h * w * 64 * 2
    wbytes = 9 * 64 * 64 * 2
    flop = 2.0 * 64 * 64 * 9 * h * w * B
    shape_key = "n%d_%dx%d" % (B, h, w)

    def line(kname, with_res):
        ms = [a.elapsed_time(b) for a, b, r in evs if r == with_res]
        if not ms:
            return None
        mean_ms = sum(ms) / len(ms)
        alg_bytes = (3 if with_res else 2) * tensor_bytes + wbytes        # in + out (+ residual) + weights
        ach = alg_bytes / (mean_ms * 1e-3) / 1e9
        traffic, rec = pmc_traffic(kname, shape_key)
        return {"kernel": kname, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                "traffic": traffic, "traffic_source": rec and rec.get("source"), "launches_timed": len(ms), "mean_launch_ms": round(mean_ms, 4),
                "bytes_per_launch": alg_bytes, "mfma_tflops": round(flop / (mean_ms * 1e-3) / 1e12, 1),
                "mfma_frac": round(flop / (mean_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
    roof = line("conv3x3_c64_bf16_v2_kernel", False)            # the launches without a residual input (conv_pre of every block)
    roof.update({"bound": "hbm", "bound_note": "priced against HBM (the nearer ceiling) with mfma_frac beside it; the counters show neither saturated "
                                               "(DESIGN.md section 8): issue / latency inside the tile pipeline at a power-limited clock",
                 "kernel": roof["kernel"] + " (64->64 3x3 trunk convolution, bf16 NHWC, folded BN + PReLU epilogue)",
                 "how": "HIP events around the eager launches of 3 passes run right after the timed graph replays",
                 "residual_variant": line("conv3x3_c64_bf16_kernel", True)})  # conv_post + Add: the v1 kernel, 1.5x the bytes
    st = pmc_step_traffic(shape_key)
    if st is not None:
        tot = st["fetch_bytes_per_step"] + st["write_bytes_per_step"]
        roof["step"] = {"hbm_bytes_per_batch": tot, "hbm_bytes_per_frame": st["bytes_per_frame"], "achieved_gbs": round(tot / (dt / args.steps) / 1e9, 1),
                        "frac_of_hbm_peak": round(tot / (dt / args.steps) / 1e9 / PEAK_HBM_GBS, 4), "source": st.get("source")}
    out = {
        "metric": "upscaled frames/s (inference, generator only) at %s" % ("256->512" if (h, w) == (256, 256) else "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w)),
        "value": round(B * args.steps / dt, 1), "unit": "frames/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "C5: make_upscaler_orig((%d,%d,3),k=3,x2,res=%d).predict, BN folded, bf16 NHWC activations, fp32 accumulate, "
                               "batch %d, one hipGraph replay per batch" % (2 * h, 2 * w, args.res_blocks, B), "global_batch": B,
                   "frame": "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w), "parallelism": "dp1", "launch": "eager" if args.no_graph else "hipGraph replay"},
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c5(args, h, w)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=["c2", "c5"], help="c2: the train step (headline); c5: inference-only generator, bf16, hipGraph")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU (default 8; 32 for --config c5)")
    ap.add_argument("--lr-size", type=int, default=256, help="low-res frame edge (output is 2x)")
    ap.add_argument("--lr-width", type=int, default=0, help="low-res frame width if not square (e.g. --lr-size 540 --lr-width 960: config C4's frames)")
    ap.add_argument("--res-blocks", type=int, default=9)
    ap.add_argument("--kernel-size", type=int, default=3, choices=[3, 5], help="residual-block / up-sampling kernel (reference default: 5)")
    ap.add_argument("--disc", default="patchgan", choices=["patchgan", "simple"])
    ap.add_argument("--disc-activation", default="none", choices=["none", "sigmoid", "log-sigm", "tanh", "bi-log"])
    ap.add_argument("--gan-losses", default="wass", choices=["wass", "rel"], help="WassersteinLosses (C2 as SURVEY 8d defines it) or RelativisticLosses('log-sigm')")
    ap.add_argument("--content", default="mse", choices=["mse", "vgg_mse"],
                    help="content loss: pixel MSE (C2 as SURVEY 8d defines it) or the reference's default VGG_MSE_LOSS form with "
                         "seeded random VGG19 weights (ImageNet weights cannot be fetched offline)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16: activations of G and D in bf16 (C3/C4's arithmetic)")
    ap.add_argument("--trunk-dtype", default=None, choices=["fp32", "bf16", "bf16+tail"],
                    help="partial mixed precision (generator only); superseded by --dtype bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one hipGraph per step")
    ap.add_argument("--fused-step", action="store_true",
                    help="secondary line, never the headline: the documented fast path without the reference's separate predict-mode "
                         "generator pass (train_gan3.py:346) -- the critic trains on the fakes of the generator's one training-mode forward")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    args = ap.parse_args()
    if args.config == "c5":
        if args.gpus != 1:
            raise SystemExit("--config c5 is a single-GPU configuration")
        return run_c5(args)
    args.batch = args.batch or 8

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
    if world == 1 and os.environ.get("VCG_BENCH_FORCE_GROUP") == "1":
        # self-check of the N > 1 code path on one GPU over a REAL (1-rank) RCCL communicator: the multi-graph plan, the asynchronous bucket
        # all-reduce, the per-collective events and the per-rank gather all run; the record says so (its number is a 1-GPU number)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
        group = dist.group.WORLD

    h = args.lr_size
    w = args.lr_width or h
    k = args.kernel_size
    bf16 = args.dtype == "bf16"
    trunk_dtype = args.trunk_dtype or ("bf16+tail" if bf16 else "fp32")
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=k, upscale_factor=2, res_block_num=args.res_blocks, seed=7,
                              trunk_dtype=trunk_dtype)
    dkw = {"dtype": "bf16"} if bf16 else {}
    D = (PM.make_discriminator_patchgan_70((2 * h, 2 * w, 3), args.disc_activation, seed=11, **dkw) if args.disc == "patchgan"
         else PM.make_discriminator_simple_512((2 * h, 2 * w, 3), args.disc_activation, seed=11, **dkw))
    if group is not None:                     # identical replicas: broadcast rank 0's weights
        for m in (G, D):
            _dist.broadcast_(m.ps.params, group)
            _dist.broadcast_(m.ps.state, group)
            m.refresh()
    content = "mse" if args.content == "mse" else PM.VGG_MSE_LOSS((2 * h, 2 * w, 3), 0.1, vgg19="random").loss
    fac = (lambda: PM.WassersteinLosses()) if args.gan_losses == "wass" else (lambda: PM.RelativisticLosses(loss_activation="log-sigm"))
    gen_train, disc_train, gan_train = PM.make_and_compile_gan2(
        G, D, (h, w, 3), (2 * h, 2 * w, 3), content, 1.0, fac, 1e-5, optimizer=PM.Adam(), process_group=group,
        fused_step=args.fused_step)
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

    # One hipGraph per step (no host read inside the step, for any of the reference's losses).  Under DP the step is cut at
    # its collectives into several graphs, with the RCCL all-reduces issued eagerly between the replays.
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
    bf16_trunk = trunk_dtype != "fp32" and k == 3
    dom_tags = ("trunk_conv", "trunk_conv_dgrad")
    res_tags = ("trunk_conv_res", "trunk_conv_dgrad_res")     # bf16: the launches with a residual input run the v1 kernel
    if not use_graph:
        rt.prof = KernelProf(dom_tags + res_tags)
    if group is not None:
        trainer.coll_prof = {}                  # HIP events around every collective (compute stream: the exposed time)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        losses = step(lr, hr)
        marks[i + 1].record()                   # per-step device time of this rank, read after the timed region (no sync inside it)
    sync()
    dt = time.perf_counter() - t0
    prof, rt.prof = rt.prof, None
    coll_prof, trainer.coll_prof = trainer.coll_prof, None
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    roof_note = "HIP events around every launch of the kernel inside the timed region"
    if use_graph:
        # kernels inside a graph replay cannot be bracketed one by one: time the same kernel launches with HIP
        # events in eager steps of the same workload right after the timed region
        rt.prof = prof = KernelProf(dom_tags + res_tags)
        for _ in range(3):
            trainer.train_step(lr, hr)
        torch.cuda.synchronize()
        rt.prof = None
        roof_note = ("the timed region replays the recorded hipGraph(s) of the step, which cannot be bracketed per kernel; these are HIP-event "
                     "timings of the same launches in 3 eager steps of the same workload run right after it (a different clock / "
                     "thermal state than the replays: the rocprofv3 --stats average of the replayed launches is under profiles/)")

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else rt.device)
    dp_info = None
    if group is not None:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX, group=group)
        # what makes an N > 1 record explain itself: every rank's step times and the time its compute stream waited at each collective
        mine = {"rank": rank, "wall_s": round(dt, 4), "step_ms_min": round(step_ms[0], 3), "step_ms_median": round(step_ms[len(step_ms) // 2], 3),
                "step_ms_max": round(step_ms[-1], 3),
                "collectives_ms": {name: round(sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1), 4) for name, ev in sorted((coll_prof or {}).items())}}
        allr = [None] * world
        dist.all_gather_object(allr, mine, group=group)
        names = sorted(allr[0]["collectives_ms"])
        dp_info = {"rccl_ranks": dist.get_world_size(group), "backend": dist.get_backend(group),
                   "bucket_bytes": {"d_bucket": int(D.ps.n_trainable) * 4, "g_bucket": int(G.ps.n_trainable) * 4,
                                    "means": 8 if trainer.relativistic else 0},
                   "collective_ms_mean_over_ranks": {nm: round(sum(r["collectives_ms"][nm] for r in allr) / world, 4) for nm in names},
                   "collective_ms_max_over_ranks": {nm: round(max(r["collectives_ms"][nm] for r in allr), 4) for nm in names},
                   "collectives_note": "HIP events on the compute stream around each collective = the time the step is exposed to it; the critic's bucket "
                                       "is issued (d_bucket_issue) before the generator's training-mode forward and waited for after it (d_bucket_wait)",
                   "per_rank": allr}
    dt = float(dt_t.item())
    frames = args.batch * world * args.steps

    # roofline of the dominant kernel: the 64->64 trunk convolution (forward and data gradient run the same kernel)
    mean_ms, launches = prof.mean_ms(dom_tags)
    flop_per_launch = 2.0 * (64 * 64 * k * k) * (h * w) * args.batch           # 2 * MAC/pixel * pixels * frames
    shape_key = "n%d_%dx%d" % (args.batch, h, w)
    roof = None
    if mean_ms:
        tfl = flop_per_launch / (mean_ms * 1e-3) / 1e12
        if bf16_trunk:
            # bf16 NHWC: 142 KB of traffic per 16x32-pixel tile against 9.2k MFMA cycles -- near the ridge.  Both ratios are printed (HBM
            # against the 8 TB/s spec, MFMA against the 2.5 PFLOP/s dense spec); `bound` names what the PMC passes show for this kernel
            # (DESIGN.md section 8): neither pipe is saturated -- issue / latency inside the tile pipeline, at a power-limited clock.
            kname = "conv3x3_c64_bf16_v2_kernel"        # the launches without a residual input; those with one run conv3x3_c64_bf16_kernel (v1), below
            alg_bytes = 2 * args.batch * h * w * 64 * 2 + 9 * 64 * 64 * 2
            ach = alg_bytes / (mean_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "bound_note": "priced against HBM (the nearer of the two ceilings: frac) with mfma_frac beside it; the counters "
                                                  "(profiles/, MfmaUtil and achieved GB/s) show neither saturated: the limiter is issue/latency in the tile pipeline",
                    "kernel": kname + " (64->64 3x3 trunk conv on bf16 NHWC without a residual input, forward + dgrad)", "achieved": round(ach, 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "bytes_per_launch": alg_bytes,
                    "mfma_tflops": round(tfl, 1), "mfma_frac": round(tfl / PEAK_BF16_MFMA_TFLOPS, 4)}
            r_ms, r_n = prof.mean_ms(res_tags)
            if r_ms:
                r_bytes = 3 * args.batch * h * w * 64 * 2 + 9 * 64 * 64 * 2          # input + residual + output + weights
                r_traffic, _ = pmc_traffic("conv3x3_c64_bf16_kernel", shape_key)
                roof["residual_variant"] = {"kernel": "conv3x3_c64_bf16_kernel (the same convolution with a residual / skip-gradient input)",
                                            "launches_timed": r_n, "mean_launch_ms": round(r_ms, 4), "bytes_per_launch": r_bytes,
                                            "achieved": round(r_bytes / (r_ms * 1e-3) / 1e9, 1), "frac": round(r_bytes / (r_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                            "mfma_tflops": round(flop_per_launch / (r_ms * 1e-3) / 1e12, 1), "traffic": r_traffic}
        else:
            kname = "conv_fwd_kernel<%d, %d, 1, 8, %d>" % (k, k, 2 if k == 3 else 1)
            roof = {"bound": "mfma", "kernel": kname + " (64->64 %dx%d trunk conv, forward + dgrad)" % (k, k),
                    "achieved": round(tfl, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tfl / PEAK_F32_MFMA_TFLOPS, 4), "flop_per_launch": flop_per_launch}
        traffic, rec = pmc_traffic(kname, shape_key)
        st = pmc_step_traffic(shape_key) if bf16 else None
        if st is not None:
            # the whole step against HBM: measured bytes of every launch of a step (PMC) / the step's wall time
            step_s = dt / args.steps
            roof["step"] = {"hbm_bytes_per_step": st["fetch_bytes_per_step"] + st["write_bytes_per_step"], "hbm_bytes_per_frame": st["bytes_per_frame"],
                            "algorithmic_gb_per_frame": st.get("algorithmic_gb_per_frame"),
                            "achieved_gbs": round((st["fetch_bytes_per_step"] + st["write_bytes_per_step"]) / step_s / 1e9, 1),
                            "frac_of_hbm_peak": round((st["fetch_bytes_per_step"] + st["write_bytes_per_step"]) / step_s / 1e9 / PEAK_HBM_GBS, 4),
                            "source": st.get("source")}
        roof.update({"traffic": traffic, "traffic_source": rec and rec.get("source"),
                     "algorithmic_bytes": roof.get("bytes_per_launch", 2 * args.batch * h * w * 64 * 4 + k * k * 64 * 64 * 4),
                     "launches_timed": launches, "mean_launch_ms": round(mean_ms, 4), "how": roof_note})

    if rank == 0:
        dts = "f32" if trunk_dtype == "fp32" else ("bf16" if bf16 else "generator %s activations bf16 / f32 elsewhere (mixed; not C2)"
                                                   % ("trunk" if trunk_dtype == "bf16" else "trunk + up-sampling + final conv"))
        wl = "C2" if (h, w) == (256, 256) and not bf16 else "C3 per-GPU shard" if (h, w) == (256, 256) else "C4 frame size" if (h, w) == (540, 960) else "custom size"
        out = {
            "metric": METRIC if (h, w) == (256, 256) else METRIC.replace("256->512", "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w)), "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dts, "data": "synthetic",
            "config": {"workload": wl + ": make_upscaler_orig((%d,%d,3),k=%d,x2,res=%d) + %s%s, batch %d/GPU, "
                                   "gan2 wiring, %s + %s, %s"
                                   % (2 * h, 2 * w, k, args.res_blocks, "PatchGAN-70" if args.disc == "patchgan" else "simple_512",
                                      "" if args.disc_activation == "none" else "(%s)" % args.disc_activation, args.batch,
                                      "Wasserstein" if args.gan_losses == "wass" else "Relativistic(log-sigm)",
                                      "pixel-MSE" if args.content == "mse" else "VGG_MSE_LOSS(0.1), random VGG19 weights",
                                      "FUSED step (extension, not the headline): no separate predict pass, 3 G + 9 D forward-equivalents" if args.fused_step
                                      else "faithful 3-call step incl. predict pass"),
                       "global_batch": args.batch * world, "frame": "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w),
                       "parallelism": "dp%d" % world, "launch": ("hipGraph replay" if group is None else "%d hipGraphs per step around the RCCL all-reduces" % len(trainer._graph)) if use_graph else "eager"},
            "last_losses": [round(float(v), 6) for v in losses],
            "roofline": roof,
        }
        if dp_info is not None:
            out["data_parallel"] = dp_info
            if world == 1:
                out["rehearsal"] = "VCG_BENCH_FORCE_GROUP=1: the data-parallel code path over a 1-rank RCCL communicator on one GPU (self-check, not a scaling measurement)"
            if rehearsal:
                out["rehearsal"] = "VCG_BENCH_REHEARSAL=1: all ranks on ONE GPU, gloo through the host instead of RCCL -- exercises the launcher / multi-graph / reduction logic; its numbers are NOT a scaling measurement"
        if world == 1 and not args.no_cpu_baseline and args.content == "mse" and w == h:
            out["cpu_baseline"] = cpu_baseline(args, h, w)
        print(json.dumps(out), flush=True)
    if group is not None:
        dist.barrier(group=group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
