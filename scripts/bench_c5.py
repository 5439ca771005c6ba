"""Config C5 of BASELINE.json: inference-only generator, batch 32, 256->512, bf16 storage, one hipGraph replay per
batch, 1 GPU.  Prints one JSON line in bench.py's format (this is a secondary configuration: the driver's headline
run is bench.py / C2).  python scripts/bench_c5.py [--batch 32] [--steps 20] [--warmup 5]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-cycle_gan-upscaling_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lr-size", type=int, default=256)
    ap.add_argument("--lr-width", type=int, default=0, help="non-square frames, e.g. --lr-size 540 --lr-width 960 (config C4's frame size)")
    ap.add_argument("--res-blocks", type=int, default=9)
    args = ap.parse_args()

    import torch
    from upscaler import _engine as E
    from upscaler import data as PD
    from upscaler import model as PM

    h, B = args.lr_size, args.batch
    w = args.lr_width or h
    G = PM.make_upscaler_orig((2 * h, 2 * w, 3), kernel_size=3, upscale_factor=2, res_block_num=args.res_blocks, seed=7)
    inf = G.to_inference_bf16()
    rt = E.Runtime.get()
    g1 = torch.Generator().manual_seed(1234)
    x = PD.frames_u8_to_device(torch.randint(0, 256, (B, h, w, 3), generator=g1, dtype=torch.uint8))
    inf.capture(B, h, w)
    for _ in range(args.warmup):
        inf.replay(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        inf.replay(x)
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
    ms = [a.elapsed_time(b) for a, b, _ in evs]
    mean_ms = sum(ms) / len(ms)
    tensor_bytes = B * h * w * 64 * 2
    nres = sum(1 for _, _, r in evs if r)
    alg_bytes = (2 * len(evs) + nres) / len(evs) * tensor_bytes + 9 * 64 * 64 * 2        # in + out (+ residual) + weights
    ach = alg_bytes / (mean_ms * 1e-3) / 1e9
    flop = 2.0 * 64 * 64 * 9 * h * w * B
    out = {
        "metric": "upscaled frames/s (inference, generator only) at %s" % ("256->512" if (h, w) == (256, 256) else "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w)), "value": round(B * args.steps / dt, 1), "unit": "frames/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "C5: make_upscaler_orig((%d,%d,3),k=3,x2,res=%d).predict, BN folded, bf16 NHWC activations, fp32 accumulate, "
                               "batch %d, one hipGraph replay per batch" % (2 * h, 2 * w, args.res_blocks, B), "global_batch": B,
                   "frame": "%dx%d->%dx%d" % (h, w, 2 * h, 2 * w), "launch": "hipGraph replay"},
        "roofline": {"bound": "hbm", "kernel": "conv3x3_c64_bf16_kernel (64->64 3x3 trunk convolution, bf16 NHWC)", "achieved": round(ach, 1),
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                     "launches_timed": len(evs), "mean_launch_ms": round(mean_ms, 4), "bytes_per_launch": alg_bytes,
                     "mfma_tflops": round(flop / (mean_ms * 1e-3) / 1e12, 1),
                     "how": "HIP events around the eager launches of 3 passes run right after the timed graph replays"},
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
