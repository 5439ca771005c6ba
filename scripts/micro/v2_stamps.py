"""Where a tile's cycles go in conv3x3_c64_bf16_v2_kernel: runs the diagnostic build made by scripts/micro/v2_stamps.sh
(s_memtime brackets, summed per wave) at the C5 trunk shape and prints the share of each segment.
python scripts/micro/v2_stamps.py [batch] [variant: plain|prelu|add]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))

import numpy as np
import torch

from upscaler import _lib as L


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    variant = sys.argv[2] if len(sys.argv) > 2 else "plain"
    lib = ctypes.CDLL(os.path.join(ROOT, "video-cycle_gan-upscaling_amd", "build", "libvcg_v2_stamps.so"))
    lib.vcg_conv2d_bf16_fwd.restype = ctypes.c_int
    lib.vcg_conv2d_bf16_fwd.argtypes = [ctypes.c_void_p] * 6
    lib.vcg_debug_v2_stamps.restype = ctypes.c_int
    lib.vcg_debug_v2_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    h = w = 256
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    res = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    wk = (torch.randn(9, 64, 64, device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.rand(64, device=dev) + 0.5
    sh = torch.rand(64, device=dev)
    al = torch.rand(64, device=dev)
    d = L.ConvDesc(B, 64, h, w, 64, h, w, 3, 3, 1, 1, 1)
    ep = {"plain": L.EpilogueBf16(None, None, L.ACT_NONE, 0.0, None, None),
          "prelu": L.EpilogueBf16(sc.data_ptr(), sh.data_ptr(), L.ACT_PRELU, 0.0, al.data_ptr(), None),
          "add": L.EpilogueBf16(sc.data_ptr(), sh.data_ptr(), L.ACT_NONE, 0.0, None, res.data_ptr())}[variant]
    stream = torch.cuda.current_stream().cuda_stream
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # the stamps are those of the LAST launch: many launches = the sustained state
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rc = lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wk.data_ptr(), y.data_ptr(), ctypes.byref(ep), stream)
    assert rc == 0, rc
    e0.record()
    for _ in range(iters):
        rc = lib.vcg_conv2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wk.data_ptr(), y.data_ptr(), ctypes.byref(ep), stream)
        assert rc == 0, rc
    e1.record()
    torch.cuda.synchronize()
    print("%d launches back to back: %.1f us per launch (events)" % (iters, 1e3 * e0.elapsed_time(e1) / iters))
    out = np.zeros(256 * 4 * 6, dtype=np.uint64)
    assert lib.vcg_debug_v2_stamps(out.ctypes.data) == 0
    full = out.reshape(256, 4, 6).astype(np.float64)
    s = full[:, :, :4]
    tiles = B * (h // 16) * (w // 32) / 256.0
    per = s / tiles                                       # s_memtime ticks (100 MHz constant clock) per tile
    names = ["phase A", "phase B", "vmcnt wait", "barrier"]   # A: half 0 + DMA of the next tile + drain of the previous half 1; B: half 1 + drain of half 0
    tot = per.sum(axis=2)
    print("variant %s, batch %d: %.1f tiles per workgroup; s_memtime ticks per tile (mean over 256 workgroups x 4 waves)" % (variant, B, tiles))
    for i, nm in enumerate(names):
        print("  %-11s mean %8.1f  min %8.1f  max %8.1f   share %5.1f %%" % (nm, per[:, :, i].mean(), per[:, :, i].min(), per[:, :, i].max(),
                                                                            100 * per[:, :, i].sum() / tot.sum()))
    print("  total       mean %8.1f  (per wave: %s)" % (tot.mean(), " ".join("%.1f" % v for v in tot.mean(axis=0))))
    print("  whole kernel: %.0f core clocks in %.1f us per wave (mean) -> the chip held %.3f GHz" % (
        full[:, :, 4].mean(), full[:, :, 5].mean() / 100.0, full[:, :, 4].sum() / full[:, :, 5].sum() * 0.1))


if __name__ == "__main__":
    main()
