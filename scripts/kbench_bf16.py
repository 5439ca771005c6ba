"""Micro-benchmark of the bf16-storage kernels at the C5 shapes (batch 32, 256x256 -> 512x512):
HIP-event time per launch, TFLOP/s against the dense bf16 MFMA peak and algorithmic GB/s against HBM.
python scripts/kbench_bf16.py [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))

import torch

from upscaler import _engine as E
from upscaler import _lib as L

PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rt = E.Runtime.get()
    dev = rt.device
    h = w = 256
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    res = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    wk = (torch.randn(9, 64, 64, device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.rand(64, device=dev) + 0.5
    sh = torch.rand(64, device=dev)
    al = torch.rand(64, device=dev)
    d = L.ConvDesc(B, 64, h, w, 64, h, w, 3, 3, 1, 1, 1)
    flop = 2.0 * 64 * 64 * 9 * h * w * B
    for name, ep, nbytes in (
            ("trunk 3x3 64->64 plain", L.EpilogueBf16(None, None, L.ACT_NONE, 0.0, None, None), 2 * x.numel() * 2),
            ("trunk + BN-fold + PReLU", L.EpilogueBf16(sc.data_ptr(), sh.data_ptr(), L.ACT_PRELU, 0.0, al.data_ptr(), None), 2 * x.numel() * 2),
            ("trunk + BN-fold + Add", L.EpilogueBf16(sc.data_ptr(), sh.data_ptr(), L.ACT_NONE, 0.0, None, res.data_ptr()), 3 * x.numel() * 2)):
        def run():
            L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wk.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream), name)
        ms = timeit(run)
        print("%-28s B=%d  %.3f ms  %7.1f TFLOP/s (%.1f%% of %.0f)  %6.0f GB/s algorithmic (%.1f%% of %.0f)"
              % (name, B, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / PEAK_TF, PEAK_TF, nbytes / ms / 1e6,
                 100 * nbytes / ms / 1e6 / PEAK_GBS, PEAK_GBS), flush=True)
    # the same launch on other operand statistics: the kernel is power-limited on random data, so its time follows the toggle rate
    # of the operands, not its instruction stream
    ep = L.EpilogueBf16(None, None, L.ACT_NONE, 0.0, None, None)
    for name, xin in (("  .. input relu(randn)", torch.relu(x.float()).to(torch.bfloat16)), ("  .. input zeros", torch.zeros_like(x))):
        def run():
            L.check(rt.lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), xin.data_ptr(), wk.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream), name)
        ms = timeit(run)
        print("%-28s B=%d  %.3f ms  %7.1f TFLOP/s" % (name, B, ms, flop / ms / 1e9), flush=True)


def bench_convt(B):
    rt = E.Runtime.get()
    dev = rt.device
    h = w = 256
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    y = torch.empty(B, 2 * h, 2 * w, 256, dtype=torch.bfloat16, device=dev)
    wk = (torch.randn(9, 256, 64, device=dev) * 0.05).to(torch.bfloat16)
    d = L.ConvDesc(B, 64, h, w, 256, 2 * h, 2 * w, 3, 3, 2, 0, 0)
    ep = L.EpilogueBf16(None, None, L.ACT_LRELU, 0.2, None, None)
    flop = 2.0 * 64 * 256 * 9 * h * w * B
    nbytes = (x.numel() + y.numel()) * 2

    def run():
        L.check(rt.lib.vcg_conv_transpose2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wk.data_ptr(), y.data_ptr(), ctypes.byref(ep), rt.stream), "convT")
    ms = timeit(run)
    print("%-28s B=%d  %.3f ms  %7.1f TFLOP/s (%.1f%% of %.0f)  %6.0f GB/s algorithmic (%.1f%% of %.0f)"
          % ("convT 3x3 s2 64->256 +LReLU", B, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / PEAK_TF, PEAK_TF, nbytes / ms / 1e6,
             100 * nbytes / ms / 1e6 / PEAK_GBS, PEAK_GBS), flush=True)
    # the same layer as four phase launches of the LDS-tiled generic kernel
    wsrc = torch.randn(3, 3, 256, 64, device=dev) * 0.05
    wfr = torch.empty(9 * 256 * 64, dtype=torch.bfloat16, device=dev)
    L.check(rt.lib.vcg_pack_conv_frag_bf16(wsrc.data_ptr(), 9, 256, 64, 1, wfr.data_ptr(), rt.stream), "pack")
    bias = torch.zeros(256, device=dev)

    def run2():
        L.check(rt.lib.vcg_conv_transpose2d_nhwc_bf16_fwd(ctypes.byref(d), x.data_ptr(), wfr.data_ptr(), bias.data_ptr(), L.ACT_LRELU, 0.2, y.data_ptr(),
                                                          rt.stream), "convT generic")
    if y.numel() * 2 <= 0xFFFFFFE0:                     # the generic kernels address a whole tensor through one buffer descriptor (< 4 GiB)
        ms = timeit(run2)
        print("%-28s B=%d  %.3f ms  %7.1f TFLOP/s (%.1f%% of %.0f)  %6.0f GB/s algorithmic (%.1f%% of %.0f)"
              % ("  .. as 4 generic phases", B, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / PEAK_TF, PEAK_TF, nbytes / ms / 1e6,
                 100 * nbytes / ms / 1e6 / PEAK_GBS, PEAK_GBS), flush=True)


def bench_final(B):
    rt = E.Runtime.get()
    dev = rt.device
    h = w = 512
    x = torch.randn(B, h, w, 256, device=dev).to(torch.bfloat16)
    y = torch.empty(B, 3, h, w, dtype=torch.float32, device=dev)
    wk = torch.randn(9, 9, 256, 3, device=dev) * 0.01
    wf = torch.empty(L.FINAL9X9_WFRAG_BYTES, dtype=torch.uint8, device=dev)
    L.check(rt.lib.vcg_pack_final9x9_bf16(wk.data_ptr(), wf.data_ptr(), rt.stream), "pack9")
    bias = torch.zeros(3, device=dev)
    d = L.ConvDesc(B, 256, h, w, 3, h, w, 9, 9, 1, 4, 4)
    flop = 2.0 * 256 * 3 * 81 * h * w * B
    nbytes = x.numel() * 2 + y.numel() * 4

    def run():
        L.check(rt.lib.vcg_conv9x9_to3_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), 1, y.data_ptr(), rt.stream), "final")
    ms = timeit(run)
    print("%-28s B=%d  %.3f ms  %7.1f TFLOP/s useful (%.1f%% of %.0f)  %6.0f GB/s algorithmic (%.1f%% of %.0f)"
          % ("final 9x9 256->3 +tanh", B, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / PEAK_TF, PEAK_TF, nbytes / ms / 1e6,
             100 * nbytes / ms / 1e6 / PEAK_GBS, PEAK_GBS), flush=True)


def bench_wgrad(B):
    rt = E.Runtime.get()
    dev = rt.device
    h = w = 256
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    dw = torch.empty(3, 3, 64, 64, device=dev)
    db = torch.empty(64, device=dev)
    d = L.ConvDesc(B, 64, h, w, 64, h, w, 3, 3, 1, 1, 1)
    ws, wsn = rt.workspace(rt.lib.vcg_conv2d_bf16_wgrad_workspace_bytes(ctypes.byref(d)))
    flop = 2.0 * 64 * 64 * 9 * h * w * B
    nbytes = (x.numel() + dy.numel()) * 2
    for name, dbp in (("trunk wgrad 3x3 64->64", None), ("trunk wgrad + dbias", db.data_ptr())):
        def run():
            L.check(rt.lib.vcg_conv2d_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), dbp, ws, wsn, rt.stream), "wgrad")
        ms = timeit(run)
        print("%-28s B=%d  %.3f ms  %7.1f TFLOP/s (%.1f%% of %.0f)  %6.0f GB/s algorithmic (%.1f%% of %.0f)"
              % (name, B, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / PEAK_TF, PEAK_TF, nbytes / ms / 1e6,
                 100 * nbytes / ms / 1e6 / PEAK_GBS, PEAK_GBS), flush=True)


def bench_norm(B):
    """BatchNormalization + PReLU of a residual block on bf16 NHWC: statistics pass, apply pass, backward (two reduction passes + apply)"""
    rt = E.Runtime.get()
    dev = rt.device
    h = w = 256
    n_el = B * h * w * 64
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, var = torch.empty(64, device=dev), torch.empty(64, device=dev)
    inv = torch.rand(64, device=dev) + 0.5
    ga, be, al = torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) - 0.5, torch.rand(64, device=dev) * 0.3
    dg, db, da = torch.empty(64, device=dev), torch.empty(64, device=dev), torch.empty(64, device=dev)
    ws, wsn = rt.workspace(max(rt.lib.vcg_norm_stats_bf16_workspace_bytes(B, 64, h * w, L.NORM_BATCH),
                               rt.lib.vcg_norm_act_bwd_bf16_workspace_bytes(B, 64, h * w, L.NORM_BATCH)))

    def stats():
        L.check(rt.lib.vcg_norm_stats_bf16(x.data_ptr(), B, 64, h * w, L.NORM_BATCH, mean.data_ptr(), var.data_ptr(), ws, wsn, rt.stream), "stats")

    def fwd():
        L.check(rt.lib.vcg_norm_act_fwd_bf16(x.data_ptr(), B, 64, h * w, ga.data_ptr(), be.data_ptr(), 0, L.ACT_PRELU, 0.0, al.data_ptr(), None,
                                             y.data_ptr(), rt.stream), "fwd")

    def bwd():
        L.check(rt.lib.vcg_norm_act_bwd_bf16(x.data_ptr(), dy.data_ptr(), B, 64, h * w, L.NORM_BATCH, mean.data_ptr(), inv.data_ptr(), ga.data_ptr(),
                                             be.data_ptr(), L.ACT_PRELU, 0.0, al.data_ptr(), 1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                             da.data_ptr(), ws, wsn, rt.stream), "bwd")
    stats()
    for name, fn, passes in (("BN statistics (1 read)", stats, 1), ("BN + PReLU apply (r + w)", fwd, 2), ("BN + PReLU backward (2r + 2r + w)", bwd, 5)):
        ms = timeit(fn)
        print("%-34s B=%d  %.3f ms  %6.0f GB/s algorithmic (%.1f%% of %.0f)" % (name, B, ms, passes * n_el * 2 / ms / 1e6,
                                                                                100 * passes * n_el * 2 / ms / 1e6 / PEAK_GBS, PEAK_GBS), flush=True)


if __name__ == "__main__":
    main()
    bench_norm(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    bench_convt(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    bench_final(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    bench_wgrad(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
