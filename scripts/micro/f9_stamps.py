"""Where an input row's cycles go in conv9x9_c256to3_bf16_kernel (the bf16 final/conv): runs the diagnostic build made by
scripts/micro/f9_stamps.sh and prints the share of each segment.  python scripts/micro/f9_stamps.py [batch] [h] [w] [launches]"""
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
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    lib = ctypes.CDLL(os.path.join(ROOT, "video-cycle_gan-upscaling_amd", "build", "libvcg_f9_stamps.so"))
    for n, a in (("vcg_conv9x9_to3_bf16_fwd", [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
                 ("vcg_pack_final9x9_bf16", [ctypes.c_void_p] * 3), ("vcg_debug_f9_stamps", [ctypes.c_void_p])):
        getattr(lib, n).restype = ctypes.c_int
        getattr(lib, n).argtypes = a
    dev = torch.device("cuda:0")
    x = torch.randn(B, h, w, 256, device=dev).to(torch.bfloat16)
    wk = torch.randn(9, 9, 256, 3, device=dev) * 0.01
    wf = torch.empty(L.FINAL9X9_WFRAG_BYTES, dtype=torch.uint8, device=dev)
    y = torch.empty(B, 3, h, w, device=dev)
    bias = torch.zeros(3, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vcg_pack_final9x9_bf16(wk.data_ptr(), wf.data_ptr(), st) == 0
    d = L.ConvDesc(B, 256, h, w, 3, h, w, 9, 9, 1, 4, 4)
    run = lambda: lib.vcg_conv9x9_to3_bf16_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), 1, y.data_ptr(), st)
    assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        assert run() == 0
    e1.record()
    torch.cuda.synchronize()
    print("batch %d %dx%d: %.1f us per launch (events, %d launches)" % (B, h, w, 1e3 * e0.elapsed_time(e1) / iters, iters))
    out = np.zeros(512 * 4 * 8, dtype=np.uint64)
    assert lib.vcg_debug_f9_stamps(out.ctypes.data) == 0
    full = out.reshape(512, 4, 8).astype(np.float64)
    full = full[full[:, 0, 5] > 0]
    rows = full[:, :, 5]
    names = ["vmcnt wait", "DMA issue", "MFMA loop", "partial+barrier", "combine+store", None, None, "acc shift"]
    tot = full[:, :, :5].sum() + full[:, :, 7].sum()
    print("%d workgroups, %.1f input rows per wave; s_memtime ticks per row (mean / min / max over waves)" % (full.shape[0], rows.mean()))
    for i, nm in enumerate(names):
        if nm is None:
            continue
        per = full[:, :, i] / rows
        print("  %-16s %8.1f %8.1f %8.1f   share %5.1f %%" % (nm, per.mean(), per.min(), per.max(), 100 * full[:, :, i].sum() / tot))
    print("  total per row    %8.1f" % (tot / rows.sum()))
    print("  whole kernel: %.0f core clocks per wave (mean)" % full[:, :, 6].mean())


if __name__ == "__main__":
    main()
