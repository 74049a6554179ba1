#!/bin/bash
# Runs ON THE GPU BOX: per-dispatch durations of k_fast (the two passes of a batch alternate) from a rocprofv3 kernel trace of the headline step.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_fast
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --headline-only --streams 1 > $OUT/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fast" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size") for r in rows]
print("k_fast dispatches (us):", [round(x, 1) for x in d[-12:]])
PY
