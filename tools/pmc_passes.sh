#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats and the counter passes of ANY command, every pass its own run (a --pmc pass is never combined
# with other tracing domains; MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  CSV output under gpurun_out/pmc_<TAG>/<pass>/.
#   tools/pmc_passes.sh lm -- python tools/exp.py lm --reps 1
# Back in the build container: python tools/pmc_summary.py lm [kernel name prefixes]   -> per-kernel means (copy what is to be judged to profiles/).
# The headline's own profile (profiles/rNN_{kernel_stats,pmc,sq}.csv, pmc_latest.json): tools/gpu_profile.sh + tools/summarize_prof.py.
TAG=$1; shift; [ "$1" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout -k 5 ${PMC_TIMEOUT:-240} rocprofv3 "$@" --output-format csv -d $OUT/$n -- $CMD > $OUT/$n.log 2>&1; }
CMD="$*"
pass stats --kernel-trace --stats
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pass sq3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
pass tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
find $OUT -name "*.csv" | head -30; tail -2 $OUT/tcc.log | cut -c1-300
