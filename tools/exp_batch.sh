#!/bin/bash
# Runs ON THE GPU BOX: times -D variants at several batch sizes.  usage: tools/exp_batch.sh "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" "$@"; do
  out=/tmp/liborbhip_exp.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -I$R/include $R/awesome-orb-slam3-3dvisioncraft-version_amd/csrc/*.hip -o $out $v 2>/dev/null || { echo "build failed: $v"; continue; }
  for b in 64 512; do
    ORBHIP_LIB=$out python bench.py --headline-only --no-cpu-baseline --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] batch $b', d['value'], d['kernel_ms'])"
  done
done
