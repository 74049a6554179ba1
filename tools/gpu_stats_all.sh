#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the full bench (all legs).  -> gpurun_out/prof_<tag>/stats_all
TAG=${1:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats_all -o trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/stats_all.log 2>&1
python3 - <<PY
import sqlite3
c = sqlite3.connect("$OUT/stats_all/trace_results.db")
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-44s calls=%-5d total_us=%-12.1f avg_us=%-10.2f %.2f%%" % (r[0][:44], r[1], r[2], r[3], r[4]))
PY
