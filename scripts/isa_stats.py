"""Per-kernel ISA statistics (branches, waits, loads, MFMAs, registers) of a csrc/*.hip file -- a quick
static look at what hipcc made of a kernel.  Usage: python scripts/isa_stats.py conv_fwd [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-cycle_gan-upscaling_amd", "csrc")


def main():
    name = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    out = "/tmp/%s.s" % name
    subprocess.run(["/opt/rocm/bin/hipcc", "-S", os.path.join(CSRC, name + ".hip"), "-o", out, "-O3", "--offload-arch=gfx950",
                    "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only"], check=True,
                   stderr=subprocess.DEVNULL)
    s = open(out).read()
    parts = re.split(r"\n\t\.type\t(_Z\w+),@function\n", s)
    meta = dict(re.findall(r"\.name:\s+(_Z\w+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", s))
    for i in range(1, len(parts), 2):
        nm, body = parts[i], parts[i + 1].split(".end_amdhsa_kernel")[0]
        if flt and flt not in nm:
            continue
        c = lambda pat: len(re.findall(pat, body))
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", parts[i + 1])
        ac = re.search(r"\.amdhsa_accum_offset (\d+)", parts[i + 1])
        print("%-58s vgpr+agpr=%s accum_off=%s cbranch=%d vmcnt=%d gload=%d gstore=%d mfma=%d ds_read=%d ds_write=%d valu~%d"
              % (nm[14:72], vg.group(1) if vg else "?", ac.group(1) if ac else "?", c(r"s_cbranch"), c(r"vmcnt"), c(r"global_load"),
                 c(r"global_store"), c(r"v_mfma"), c(r"ds_read"), c(r"ds_write"), c(r"\n\tv_(?!mfma)")))


if __name__ == "__main__":
    main()
