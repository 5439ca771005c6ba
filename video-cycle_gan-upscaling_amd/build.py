"""Build libvcg_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvcg_hip.so")
SOURCES = ["conv_fwd.hip", "conv_rowchain.hip", "conv_transpose.hip", "conv_wgrad.hip", "norm.hip", "elementwise.hip", "dense.hip", "bf16_conv.hip", "bf16_norm.hip", "bf16_wgrad.hip", "bf16_wgrad9.hip", "bf16_gconv.hip", "bf16_gwgrad.hip", "bf16_head.hip", "bf16_wgrad3.hip", "api.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wno-unused-value", "-Wno-c++20-extensions"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "vcg_common.hpp"), os.path.join(ROOT, "include", "vcg.h")]
    objs, jobs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, "-c", src, "-o", obj] + FLAGS)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
