"""Where a tile's cycles go in convt3x3_c64_bf16_kernel (the bf16 up-sampling block's transposed convolution, 64 -> 256 channels): runs the
diagnostic builds made by scripts/micro/ct_stamps.sh.  python scripts/micro/ct_stamps.py [batch] [h] [w] [lib ...]"""
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
    for n, a in (("vcg_conv_transpose2d_bf16_fwd", [P, P, P, P, P, P]), ("vcg_pack_conv_kernel_bf16", [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P]),
                 ("vcg_debug_ct_stamps", [P])):
        getattr(lib, n).restype = ctypes.c_int
        getattr(lib, n).argtypes = a
    dev = torch.device("cuda:0")
    x = torch.randn(B, h, w, 64, device=dev).to(torch.bfloat16)
    wk = torch.randn(3, 3, 256, 64, device=dev) * 0.05
    wp = torch.empty(9, 256, 64, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(256, device=dev)
    y = torch.empty(B, 2 * h, 2 * w, 256, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vcg_pack_conv_kernel_bf16(wk.data_ptr(), 9, 256, 64, 0, 0, wp.data_ptr(), st) == 0
    d = L.ConvDesc(B, 64, h, w, 256, 2 * h, 2 * w, 3, 3, 2, 0, 0)
    ep = L.EpilogueBf16(None, bias.data_ptr(), L.ACT_LRELU, 0.2, None, None)
    go = lambda: lib.vcg_conv_transpose2d_bf16_fwd(ctypes.byref(d), x.data_ptr(), wp.data_ptr(), y.data_ptr(), ctypes.byref(ep), st)
    assert go() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        assert go() == 0
    e1.record()
    torch.cuda.synchronize()
    us = 1e2 * e0.elapsed_time(e1)
    out = np.zeros(256 * 8 * 5, dtype=np.uint64)
    assert lib.vcg_debug_ct_stamps(out.ctypes.data) == 0
    f = out.reshape(256, 8, 5).astype(np.float64)[:, :6, :]
    f = f[f[:, :, 3] > 0]
    t = f[:, 3]
    print("%-40s batch %d %dx%d: %.1f us per launch (%.2f TB/s written); %.1f tiles per wave; ticks per tile: body %.0f  barrier A %.0f  barrier B %.0f  (kernel %.0f)"
          % (libname, B, h, w, us, B * 4 * h * w * 512 / us / 1e6, t.mean(), (f[:, 0] / t).mean(), (f[:, 1] / t).mean(), (f[:, 2] / t).mean(), (f[:, 4] / t).mean()))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    for libname in (sys.argv[4:] or ["libvcg_ct_stamps.so"]):
        run(libname, B, h, w)
