#!/bin/bash
# Runs ON THE GPU BOX: SQ counters + kernel stats of the LBA linearisation kernels at 256 windows per launch (tools/exp_lba_windows.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/lba_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python $R/tools/exp_lba_windows.py > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -o p -- python $R/tools/exp_lba_windows.py > $OUT/sq1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $R/tools/exp_lba_windows.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $R/tools/exp_lba_windows.py > $OUT/write.log 2>&1
python - <<PY
import csv, glob, collections
def load(pat):
    f = glob.glob("$OUT/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
st = load("stats/**/*kernel_stats.csv")
for r in st:
    if "k_lba" in r["Name"]: print(r["Name"][:40], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "max us", round(float(r["MaxNs"]) / 1e3, 1))
for d in ("sq1", "fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in load(d + "/**/*counter_collection.csv"):
        if "k_lba" in r["Kernel_Name"]: agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, {c: "%.4g" % max(x) for c, x in v.items()})   # max = the 256-window launches
PY
