#!/bin/bash
# Runs ON THE GPU BOX: builds a -DFAST_PROF variant and prints the per-stage wave-cycle totals of k_fast (experiment only).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=/tmp/liborbhip_prof.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$R/include $R/awesome-orb-slam3-3dvisioncraft-version_amd/csrc/*.hip -o $out -DFAST_PROF 2>&1 | grep -E "error" 
ORBHIP_LIB=$out python - <<'PY'
import ctypes, json, subprocess, sys, os
sys.argv=['bench.py','--headline-only','--no-cpu-baseline','--steps','5','--warmup','1']
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
from orbhip import _lib
L=_lib.load()
buf=(ctypes.c_ulonglong*16)()
print('rc',L.orbx_debug_prof(buf))
v=list(buf)
names=['stage','sync0','s1_pretest','s1_compact','s2_ring','s3_score','sync1','nms_emit','out']
tot=sum(v[:9])
for n,x in zip(names,v[:9]): print('%-12s %14d  %5.1f%%'%(n,x,100.0*x/tot))
print('n1 total',v[10],'n2 total',v[11],'waves',v[12])
PY
