"""Summarise gpurun_out/pmc_<filter>/p*/k_counter_collection.csv per kernel (mean per dispatch)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "p*", "k_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, cs in agg.items():
    n = len(next(iter(cs.values())))
    if n < 3 or max(sum(v) / len(v) for v in cs.values()) < 1e5:
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    us = sum(dur[name]) / len(dur[name])
    line = "%-44s n=%3d %8.1f us" % (name[-44:], n, us)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        line += "  MfmaUtil=%.1f%% clk=%.2fGHz" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), m["GRBM_GUI_ACTIVE"] / 8 / us / 1e3)
    if "SQ_WAVE_CYCLES" in m:
        line += "  wait_any=%.0f%% wait_inst=%.0f%% active=%.0f%%" % (100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"])
    if "SQ_LDS_BANK_CONFLICT" in m:
        line += "  lds_conf=%.2g" % m["SQ_LDS_BANK_CONFLICT"]
    for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"):
        if k in m:
            line += "  %s=%.3g" % (k[9:], m[k])
    if "FETCH_SIZE" in m:
        line += "  FETCH=%.1fMB(x2 corr %.1fMB)" % (m["FETCH_SIZE"] / 1024, 2 * m["FETCH_SIZE"] / 1024)
    if "WRITE_SIZE" in m:
        line += "  WRITE=%.1fMB" % (m["WRITE_SIZE"] / 1024)
    print(line)
