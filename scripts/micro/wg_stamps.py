"""Where a tile's cycles go in wgrad3x3_c64_bf16_kernel (the bf16 trunk's weight gradient): runs the diagnostic builds made by
scripts/micro/wg_stamps.sh.  python scripts/micro/wg_stamps.py [batch] [h] [w] [lib ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))

import numpy as np
import torch

from upscaler import _lib as L


def run(libname, B, h, w):
    lib = ctypes.CDLL(os.path.join(ROOT, "video-cycle_gan-upscaling_amd", "build", libname))
    P = ctypes.c_void_p
    lib.vcg_conv2d_bf16_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.vcg_conv2d_bf16_wgrad_workspace_bytes.argtypes = [P]
    lib.vcg_conv2d_bf16_wgrad.restype = ctypes.c_int
    lib.vcg_conv2d_bf16_wgrad.argtypes = [P, P, P, P, P, P, ctypes.c_size_t, P]
    lib.vcg_debug_wg_stamps.restype = ctypes.c_int
    lib.vcg_debug_wg_stamps.argtypes = [P]
    dev = torch.device("cuda:0")
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    dw = torch.empty(3, 3, 64, 64, device=dev)
    db = torch.empty(64, device=dev)
    d = L.ConvDesc(B, 64, h, w, 64, h, w, 3, 3, 1, 1, 1)
    nws = lib.vcg_conv2d_bf16_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    go = lambda: lib.vcg_conv2d_bf16_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, st)
    assert go() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        assert go() == 0
    e1.record()
    torch.cuda.synchronize()
    us = 1e2 * e0.elapsed_time(e1)
    out = np.zeros(256 * 8 * 6, dtype=np.uint64)
    assert lib.vcg_debug_wg_stamps(out.ctypes.data) == 0
    f = out.reshape(256, 8, 6).astype(np.float64)
    print("%-36s batch %d %dx%d: %.1f us per launch incl. the reduction (%.2f TB/s read)" % (libname, B, h, w, us, B * h * w * 256 / us / 1e6))
    for name, sel in (("tap-row 0 waves", f[:, 0:2]), ("tap-row 1 waves", f[:, 2:4]), ("tap-row 2 waves", f[:, 4:6])):
        g = sel.reshape(-1, 6)
        g = g[g[:, 4] > 0]
        t = g[:, 4]
        print("  %s: %.1f tiles per wave; ticks per tile: DMA wait %.0f  barrier %.0f  DMA issue %.0f  k-steps %.0f  (kernel %.0f per tile)"
              % (name, t.mean(), (g[:, 0] / t).mean(), (g[:, 1] / t).mean(), (g[:, 2] / t).mean(), (g[:, 3] / t).mean(), (g[:, 5] / t).mean()))
    g = f[:, 6:8].reshape(-1, 6)
    g = g[g[:, 4] > 0]
    if len(g):
        t = g[:, 4]
        print("  loader waves: ticks per tile: wait for the stage %.0f  barrier %.0f  issue of the next stage %.0f  (kernel %.0f per tile)"
              % ((g[:, 0] / t).mean(), (g[:, 1] / t).mean(), (g[:, 2] / t).mean(), (g[:, 5] / t).mean()))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    for libname in (sys.argv[4:] or ["libvcg_wg_stamps.so"]):
        run(libname, B, h, w)
