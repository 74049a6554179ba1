#!/bin/bash
# Runs in the BUILD container (hipcc cross-compiles gfx950 without a GPU): builds one liborbhip variant per argument into exp_so/ so that the GPU box
# only has to time them (tools/exp_so.sh) — compiling on the box costs ~25 s of GPU-minutes per variant.
#   tools/build_variants.sh "" "-DFAST_XCD=1" "-DOCT_U=8 -DDESC_WAVES=6"
#   gpurun --timeout 200 -- 'bash tools/exp_so.sh'
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/exp_so
rm -f $R/exp_so/*.so
i=0
for v in "$@"; do
  n=$(printf "v%02d_%s" $i "$(echo "base$v" | tr -d ' ' | tr '=' '_' | tr -c 'A-Za-z0-9_\n-' '_')")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -I$R/include $R/awesome-orb-slam3-3dvisioncraft-version_amd/csrc/*.hip -o $R/exp_so/$n.so $v 2>/dev/null && echo "built exp_so/$n.so  [$v]" || echo "BUILD FAILED [$v]"
  i=$((i+1))
done
