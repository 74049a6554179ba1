#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction-mix / stall counters of the headline bench (own run, no tracing domains besides kernel dispatch).
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sq1 -o pmc -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $OUT/sq2 -o pmc -- $CMD > $OUT/sq2.log 2>&1
tail -3 $OUT/sq1.log $OUT/sq2.log
python3 - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$OUT/sq*/pmc_results.db")):
    d = sqlite3.connect(f)
    rows = list(d.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"))
    for r in rows:
        if r[0].startswith("k_"): print("%-28s %-24s n=%d mean=%.4g" % (r[0][:28], r[1], r[2], r[3]))
PY
