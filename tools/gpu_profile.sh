#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + HBM PMC passes of the default bench command.
# Usage: tools/gpu_profile.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- $CMD > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep only small summaries (gpurun_out merge is capped at 64 MiB)
find $OUT -name "*kernel_trace.csv" -size +8M -delete
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % kind, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "?").split("(")[0], r.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(r.get("Counter_Value", 0))
        with open("$OUT/%s_summary.csv" % kind, "w") as o:
            o.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
            for (k, c), (n, s) in sorted(agg.items()):
                o.write("%s,%s,%d,%.1f,%.1f\n" % (k, c, n, s, s / n))
        print(open("$OUT/%s_summary.csv" % kind).read())
PY
find $OUT -name "*counter_collection.csv" -size +4M -delete
cat $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -20 || find $OUT/stats -name "*stats*" | head
