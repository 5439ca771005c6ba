"""Where a tile's cycles go in conv_c3to64_bf16_kernel<9,3,1> as the data gradient of final/conv (3 -> 256 channels, LeakyReLU mask): runs
the diagnostic build made by scripts/micro/i9_stamps.sh.  python scripts/micro/i9_stamps.py [batch] [h] [w] [mask 0/1]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-cycle_gan-upscaling_amd"))

import numpy as np
import torch

from upscaler import _lib as L


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    mask = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    lib = ctypes.CDLL(os.path.join(ROOT, "video-cycle_gan-upscaling_amd", "build", "libvcg_i9_stamps.so"))
    P = ctypes.c_void_p
    for n, a in (("vcg_conv9x9_to3_bf16_dgrad", [P, P, P, P, ctypes.c_float, P, P]), ("vcg_pack_conv9x9_3ch_bf16", [P, ctypes.c_int, ctypes.c_int, P, P]),
                 ("vcg_debug_i9_stamps", [P])):
        getattr(lib, n).restype = ctypes.c_int
        getattr(lib, n).argtypes = a
    dev = torch.device("cuda:0")
    dy = torch.randn(B, 3, h, w, device=dev)
    wk = torch.randn(9, 9, 256, 3, device=dev) * 0.01
    wd = torch.empty(4 * L.FIRST9X9_WFRAG_BYTES, dtype=torch.uint8, device=dev)
    yprev = torch.randn(B, h, w, 256, device=dev).to(torch.bfloat16)
    dx = torch.empty(B, h, w, 256, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vcg_pack_conv9x9_3ch_bf16(wk.data_ptr(), 256, 1, wd.data_ptr(), st) == 0
    d = L.ConvDesc(B, 256, h, w, 3, h, w, 9, 9, 1, 4, 4)
    run = lambda: lib.vcg_conv9x9_to3_bf16_dgrad(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), yprev.data_ptr() if mask else None, 0.2, dx.data_ptr(), st)
    assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        assert run() == 0
    e1.record()
    torch.cuda.synchronize()
    print("batch %d %dx%d mask=%d: %.1f us per launch" % (B, h, w, mask, 1e2 * e0.elapsed_time(e1)))
    out = np.zeros(512 * 8 * 6, dtype=np.uint64)
    assert lib.vcg_debug_i9_stamps(out.ctypes.data) == 0
    full = out.reshape(512, 8, 6).astype(np.float64)
    for nm, sl, names in (("compute waves", slice(0, 6), ["MFMA loop", "barrier 1", "epilogue", "barrier 2"]),
                          ("loader waves", slice(6, 8), ["fetch issue", "barrier 1", "stash", "barrier 2"])):
        f = full[:, sl, :]
        f = f[f[:, :, 4] > 0]
        tiles = f[:, 4]
        print("%s: %.1f tiles per wave; s_memtime ticks (100 MHz) per tile, mean / min / max over waves" % (nm, tiles.mean()))
        for i, n2 in enumerate(names):
            per = f[:, i] / tiles
            print("  %-12s %8.1f %8.1f %8.1f" % (n2, per.mean(), per.min(), per.max()))
        print("  whole kernel %.0f ticks per wave = %.1f per tile" % (f[:, 5].mean(), (f[:, 5] / tiles).mean()))


if __name__ == "__main__":
    main()
