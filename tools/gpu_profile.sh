#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats, the HBM PMC passes and the SQ occupancy / issue / LDS passes of the headline bench
# command — every pass its own run (a --pmc pass is never combined with other tracing domains).
# Usage: tools/gpu_profile.sh <tag>   -> gpurun_out/prof_<tag>/{stats,fetch,write,sq1,sq2}/*.db (rocpd SQLite, ROCm 7.2) + bench.json
# Back in the build container: python tools/summarize_prof.py <tag>  -> profiles/<tag>_{kernel_stats,pmc,sq}.csv, profiles/pmc_latest.json
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only --streams 1 --no-parity-check --repeats 1"   # one stream: kernels run back to back, clean per-kernel durations
date +%T; timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- $CMD > $OUT/stats.log 2>&1
date +%T; timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
date +%T; timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- $CMD > $OUT/write.log 2>&1
date +%T; timeout -k 5 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sq1 -o pmc -- $CMD > $OUT/sq1.log 2>&1
date +%T; timeout -k 5 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $OUT/sq2 -o pmc -- $CMD > $OUT/sq2.log 2>&1
date +%T; tail -2 $OUT/*.log | cut -c1-300
ls -la $OUT/*/
