#!/bin/bash
# Runs ON THE GPU BOX: times the full-LM leg (tools/exp_lm.py) with every prebuilt variant library under exp_so/ (tools/build_variants.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in exp_so/*.so; do
  echo "$so: $(ORBHIP_LIB=$R/$so timeout 120 python tools/exp_lm.py "$@" 2>&1 | tail -1)"
done
