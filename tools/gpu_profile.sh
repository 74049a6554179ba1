#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + HBM PMC passes (separate runs) of the default bench command.
# Usage: tools/gpu_profile.sh <tag>   -> gpurun_out/prof_<tag>/{stats,fetch,write}/*.db (rocpd SQLite, ROCm 7.2)
# Back in the build container: python tools/summarize_prof.py <tag>  -> profiles/<tag>_*.csv, profiles/pmc_latest.json
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- $CMD > $OUT/write.log 2>&1
ls -la $OUT/*/
