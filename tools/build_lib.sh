#!/bin/bash
# Build the product C-ABI library for gfx950 (cross-compiles without a GPU).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/awesome-orb-slam3-3dvisioncraft-version_amd/csrc"
OUT="$ROOT/awesome-orb-slam3-3dvisioncraft-version_amd/liborbhip.so"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -ffp-contract=off + correctly rounded fp32 div/sqrt: float expressions must round exactly as the reference's
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wall -Wno-unused-function -Wno-unused-result -I"$ROOT/include" "$SRC"/*.hip -o "$OUT" "$@"
echo "built $OUT"
# the exchange entry points over RCCL (include/orbd.h): host code only, its own small library so that liborbhip.so needs no RCCL
$HIPCC -O2 -std=c++17 -fPIC -shared -Wall -I"$ROOT/include" "$SRC/orbd_exchange.cpp" -o "$ROOT/awesome-orb-slam3-3dvisioncraft-version_amd/liborbd.so" -L/opt/rocm/lib -lrccl
echo "built $ROOT/awesome-orb-slam3-3dvisioncraft-version_amd/liborbd.so"
# latency tool of the drop-in ORBextractor::operator() (mock cv:: of tests/cpp/mock_orbslam3; bench.py's extra.host_api_cv runs it on the GPU box)
mkdir -p "$ROOT/tools/bin"
g++ -std=c++17 -O2 -DORBHIP_WITH_OPENCV -I"$ROOT/tests/cpp/mock_orbslam3" -I"$ROOT/include" "$ROOT/tools/extractor_cv_latency.cpp" \
    -L"$ROOT/awesome-orb-slam3-3dvisioncraft-version_amd" -lorbhip -Wl,-rpath,'$ORIGIN/../../awesome-orb-slam3-3dvisioncraft-version_amd' -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lpthread \
    -o "$ROOT/tools/bin/extractor_cv_latency"
echo "built $ROOT/tools/bin/extractor_cv_latency"
