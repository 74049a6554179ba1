#!/bin/bash
# Runs ON THE GPU BOX: headline step vs frames per launch and stream count.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "512 3" "1024 3" "2048 3" "1024 2" "256 3"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --headline-only --steps 20 --warmup 3 --batch $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('batch $1 streams $2', 'e+m', d['value'], 'ms/step', d['ms_per_step'], 'extract', d['metric_components']['orb_extract_frames_per_s'])"
done
