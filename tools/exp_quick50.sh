#!/bin/bash
# Runs ON THE GPU BOX: like tools/exp_quick.sh with the headline's own step count (50 timed steps, 5 warm-up), twice per variant, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for so in exp_so/*.so; do
  ORBHIP_LIB=$R/$so python bench.py --no-cpu-baseline --headline-only --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$so', 'e+m', d['value'], 'extract', d['metric_components']['orb_extract_frames_per_s'], 'pyr', round(k['pyramid'],3), 'fast', round(k['fast'],3), 'desc', round(k['describe'],3))"
done; done
