#!/bin/bash
# Runs ON THE GPU BOX: like exp_variants.sh but reports the extract+match leg (kernel experiments on the matcher).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" "$@"; do
  out=/tmp/liborbhip_exp.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -I$R/include $R/awesome-orb-slam3-3dvisioncraft-version_amd/csrc/*.hip -o $out $v 2>/dev/null || { echo "build failed: $v"; continue; }
  ORBHIP_LIB=$out python bench.py --no-cpu-baseline --lm-windows 4 --lba-windows 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v]', d['extra']['extract_match']['match_only_ms'], d['extra']['extract_match']['mean_matches_per_frame'])"
done
