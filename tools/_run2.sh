R=$GRAFT_REPO_ROOT
export ORBHIP_LIB=$R/exp_so/v01_base-DLM_PAIRS_PER_EDGE_0.so
for t in 1 2 4; do timeout 200 python $R/tools/exp.py lm --threads $t --reps 3 2>&1 | tail -1; done
