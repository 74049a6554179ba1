#!/bin/bash
# Runs ON THE GPU BOX: times the extract+match leg for every prebuilt variant library under exp_so/ (built in the build container, so no GPU time goes into hipcc).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in exp_so/*.so; do
  ORBHIP_LIB=$R/$so python bench.py --no-cpu-baseline --lm-windows 4 --lba-windows 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', d['extra']['extract_match']['match_only_ms'], d['extra']['extract_match']['mean_matches_per_frame'])"
done
