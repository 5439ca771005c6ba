"""Step-level HBM traffic from the two passes of scripts/pmc_step.sh.

    python scripts/pmc_step_summary.py <gpurun_out/pmcstep_TAG> <shape key> <steps run per pass> <frames per step> <source tag>
                                       [--alg-gb-per-frame X] [--merge]

Per kernel: dispatches per step, fetch / write bytes per dispatch (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read -> doubled; WRITE_SIZE exact; both KiB -- MI355X_MICROARCH.md, HBM).  Total: sum over every dispatch of the
pass / steps (the handful of set-up kernels before the first step are included: weight initialisation, frame conversion).
--merge folds the per-kernel records into profiles/pmc_traffic.json under <shape key> (what bench.py's roofline.traffic reads) and the
step total into its "steps" list."""
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
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    d, shape, steps, frames, source = a[0], a[1], int(a[2]), int(a[3]), a[4]
    alg = None
    if "--alg-gb-per-frame" in sys.argv:
        alg = float(sys.argv[sys.argv.index("--alg-gb-per-frame") + 1])
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "p*", "k_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            n = short(r["Kernel_Name"])
            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    tot_f = tot_w = 0.0
    for n, cs in vals.items():
        fs, ws = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
        fb = 2 * 1024 * sum(fs)
        wb = 1024 * sum(ws)
        tot_f += fb
        tot_w += wb
        cnt = max(len(fs), len(ws))
        rows.append((fb + wb, n, cnt, fb / max(len(fs), 1), wb / max(len(ws), 1), sum(dur[n]) / len(dur[n])))
    rows.sort(reverse=True)
    step_bytes = (tot_f + tot_w) / steps
    print("shape %s: %d steps per pass, %d frames per step" % (shape, steps, frames))
    print("%-58s %9s %12s %12s %10s %8s" % ("kernel", "disp/step", "fetch MB/disp", "write MB/disp", "us (pmc)", "% bytes"))
    for b, n, cnt, f1, w1, us in rows[:40]:
        print("%-58s %9.1f %12.2f %12.2f %10.1f %7.2f%%" % (n[:58], cnt / steps, f1 / 1e6, w1 / 1e6, us, 100 * b / (tot_f + tot_w)))
    print("step total: fetch %.3f GB + write %.3f GB = %.3f GB per step = %.3f GB per frame"
          % (tot_f / steps / 1e9, tot_w / steps / 1e9, step_bytes / 1e9, step_bytes / frames / 1e9))
    if alg:
        print("algorithmic figure (SURVEY.md section 8d): %.2f GB per frame -> measured / algorithmic = %.2f" % (alg, step_bytes / frames / 1e9 / alg))
    if "--merge" in sys.argv:
        try:
            doc = json.load(open(OUT))
        except (OSError, ValueError):
            doc = {"note": "HBM bytes per launch from rocprofv3 PMC passes; fetch_bytes = 2 x FETCH_SIZE (gfx950), write_bytes = WRITE_SIZE", "kernels": []}
        for b, n, cnt, f1, w1, us in rows:
            if b / (tot_f + tot_w) < 0.002:
                continue
            base = n.split("<")[0] if n.startswith(("conv3x3_c64_bf16", "wgrad3x3_c64_bf16")) else n
            rec = {"kernel": base, "variant": n, "shape": shape, "fetch_bytes": int(f1), "write_bytes": int(w1), "dispatches": cnt,
                   "mean_us_under_pmc": round(us, 1), "source": source}
            doc["kernels"] = [k for k in doc["kernels"] if not (k["kernel"] == base and k["shape"] == shape and k.get("variant") == n)] + [rec]
        st = {"shape": shape, "frames_per_step": frames, "fetch_bytes_per_step": int(tot_f / steps), "write_bytes_per_step": int(tot_w / steps),
              "bytes_per_frame": int(step_bytes / frames), "algorithmic_gb_per_frame": alg, "source": source}
        doc["steps"] = [s for s in doc.get("steps", []) if s["shape"] != shape] + [st]
        json.dump(doc, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
