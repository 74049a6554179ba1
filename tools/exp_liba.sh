#!/bin/bash
# Runs ON THE GPU BOX: LocalInertialBA throughput for -D variants of the library (experiment).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" "$@"; do
  out=/tmp/liborbhip_exp.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -I$R/include $R/awesome-orb-slam3-3dvisioncraft-version_amd/csrc/*.hip -o $out $v 2>/dev/null || { echo "build failed: $v"; continue; }
  echo "variant [$v]"; ORBHIP_LIB=$out python tools/exp_inertial_batch.py 2>&1 | tail -2
done
