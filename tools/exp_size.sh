#!/bin/bash
# Runs ON THE GPU BOX: every variant library under exp_so/ on another frame shape (default 1280x720 / 1500 key points, batch 256), headline step only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in exp_so/*.so; do
  ORBHIP_LIB=$R/$so python bench.py --no-cpu-baseline --headline-only --steps 10 --warmup 2 --size ${1:-1280x720} --nfeatures ${2:-1500} --batch ${3:-256} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$so', 'e+m', d['value'], 'extract', d['metric_components']['orb_extract_frames_per_s'], {a: round(b,3) for a,b in k.items()}, 'kp', d['config']['mean_keypoints'])"
done
