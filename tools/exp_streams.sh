#!/bin/bash
# Runs ON THE GPU BOX: the headline step with one and with two HIP streams.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for st in 2 3; do
  python bench.py --no-cpu-baseline --headline-only --steps 20 --warmup 3 --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('streams $st', 'e+m', d['value'], 'ms/step', d['ms_per_step'], 'extract', d['metric_components']['orb_extract_frames_per_s'], {a: round(b,3) for a,b in k.items()}, 'matches', d['config']['mean_matches'])"
done
