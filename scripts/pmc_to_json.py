"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json, the file bench.py reads
`roofline.traffic` from (by kernel name and workload shape).

    python scripts/pmc_to_json.py <gpurun_out/pmc_dir> <shape key, e.g. n8_256x256> <source tag> [kernel substring ...]

<pmc_dir> holds p*/k_counter_collection.csv as written by scripts/pmc_kbench.sh (separate passes per counter set, as
MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes).  gfx950 correction (same guide, section HBM): FETCH_SIZE counts 64 B
per 128-B request of a wide coalesced read -> doubled; WRITE_SIZE is exact; both are reported in KiB."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


def main():
    d, shape, source = sys.argv[1], sys.argv[2], sys.argv[3]
    want = sys.argv[4:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "p*", "k_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            n = short(r["Kernel_Name"])
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    try:
        doc = json.load(open(OUT))
    except (OSError, ValueError):
        doc = {"note": "HBM bytes per launch from rocprofv3 PMC passes; fetch_bytes = 2 x FETCH_SIZE (gfx950), write_bytes = WRITE_SIZE",
               "kernels": []}
    for n, cs in sorted(agg.items()):
        if want and not any(w in n for w in want):
            continue
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        base = n.split("<")[0] if n.startswith(("conv3x3_c64_bf16", "wgrad3x3_c64_bf16")) else n      # ..._bf16_kernel (v1) / ..._bf16_v2_kernel
        rec = {"kernel": base, "variant": n, "shape": shape, "fetch_bytes": int(2 * 1024 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])),
               "write_bytes": int(1024 * sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])), "dispatches": len(cs["FETCH_SIZE"]),
               "mean_us_under_pmc": round(sum(dur[n]) / len(dur[n]), 1), "source": source}
        doc["kernels"] = [k for k in doc["kernels"] if not (k["kernel"] == rec["kernel"] and k["shape"] == shape and k.get("variant") == n)] + [rec]
        print(rec)
    json.dump(doc, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
