#!/bin/bash
# Runs ON THE GPU BOX: times every prebuilt variant library under exp_so/ (tools/build_variants.sh) with the bench's headline + extract_match legs.
# Prints per variant: extract frames/s, per-kernel ms (pyramid/fast/octree/describe), match_only_ms, mean matches per frame (a parity canary).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in exp_so/*.so; do
  ORBHIP_LIB=$R/$so python bench.py --no-cpu-baseline --lm-windows 4 --lba-windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; k=e.get('kernel_ms', d.get('kernel_ms', {}))
m=e.get('extract_match', {})
print('$so', 'extract', d['value'], 'kern', k, 'match_ms', m.get('match_only_ms'), 'matches', m.get('mean_matches_per_frame'))"
done
