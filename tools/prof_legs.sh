#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats of ALL bench legs (headline + extract_match + stereo + LBA ...), CSV output.
# Usage: tools/prof_legs.sh <tag>  -> gpurun_out/legs_<tag>/trace_kernel_stats.csv (copy to profiles/<tag>_legs_kernel_stats.csv)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/legs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --lm-windows 16 --lba-windows 4 > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
head -45 $OUT/kernel_stats.csv | cut -c1-160
