#!/bin/bash
# Runs ON THE GPU BOX: times every prebuilt variant library under exp_so/ (tools/build_variants.sh) with the headline step only.
# Prints per variant: extract+match frames/s, extract-only frames/s, per-kernel ms and the mean match count (a parity canary).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in exp_so/*.so; do
  ORBHIP_LIB=$R/$so python bench.py --no-cpu-baseline --headline-only --no-parity-check --repeats 3 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$so', 'e+m', d['value'], 'extract', d['metric_components']['orb_extract_frames_per_s'], {a: round(b,3) for a,b in k.items()}, 'matches', d['config']['mean_matches'], 'kp', d['config']['mean_keypoints'])"
done
