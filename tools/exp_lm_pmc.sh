#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (each its own run) over the LM leg (tools/exp_lm.py --reps 1); CSV output under gpurun_out/lmpmc/<pass>/.
# Back in the build container: python tools/exp_lm_pmc.py  prints per-kernel means.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/lmpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/exp_lm.py --reps 1 --windows ${1:-256}"
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/sq3 -- $CMD > $OUT/sq3.log 2>&1
timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
find $OUT -name "*.csv" | head -30; tail -2 $OUT/sq3.log
