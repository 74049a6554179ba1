#!/usr/bin/env python3
"""Per-kernel means of the passes tools/pmc_passes.sh TAG left under gpurun_out/pmc_TAG/:  python tools/pmc_summary.py TAG [kernel name prefixes]"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "pmc_" + (sys.argv[1] if len(sys.argv) > 1 else "lm"))
dur = {}
for f in glob.glob(os.path.join(base, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(base, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = [k for k in dur if not sys.argv[2:] or k.replace("void ", "").startswith(tuple(sys.argv[2:]))]
for k in want:
    if k not in acc:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print("%s  avg %.1f us x %d" % (k, dur.get(k, (0, 0))[0], dur.get(k, (0, 0))[1]))
    for n in sorted(c):
        print("    %-28s %.4g" % (n, c[n]))
    if "FETCH_SIZE" in c:
        print("    -> HBM read %.1f MB (x2 correction: %.1f), write %.1f MB per launch" % (c["FETCH_SIZE"] / 1024, c["FETCH_SIZE"] * 2 / 1024, c.get("WRITE_SIZE", 0) / 1024))
    if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
        print("    -> VALU insts/wave %.0f, LDS insts/wave %.0f, VMEM rd/wave %.0f, wave cycles/wave %.0f, wait-any share %.2f" %
              (c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_LDS"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"],
               c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"] * 4, c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
