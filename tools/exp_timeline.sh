#!/bin/bash
# Runs ON THE GPU BOX: kernel timeline of the headline step (rocprofv3 --kernel-trace, csv) for a given --streams value, and what the step's
# wall time is made of: device busy (union of kernel intervals), idle gaps, time with >= 2 kernels in flight, per-kernel wall sums.
# usage: tools/exp_timeline.sh <streams> [extra bench args]
S=${1:-3}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/timeline_s$S
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --headline-only --no-parity-check --streams $S "$@" > $OUT/run.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region of the headline = the last 10 steps of the step loop: find k_sbp_resolve_par (or k_sbp_resolve) launches, take the 10 before the extract-only leg
res = [i for i, r in enumerate(rows) if r[2].startswith("k_sbp_resolve") and "par" not in r[2]]
last = res[-1]; first = res[-11] if len(res) > 11 else res[0]
t0, t1 = rows[first][1], rows[last][1]
seg = [r for r in rows if r[0] >= t0 and r[1] <= t1]
ev = sorted([(s, 1) for s, e, _, _ in seg] + [(e, -1) for s, e, _, _ in seg])
busy = over = 0; depth = 0; prev = t0
for t, d in ev:
    if depth >= 1: busy += t - prev
    if depth >= 2: over += t - prev
    depth += d; prev = t
wall = t1 - t0
per = collections.defaultdict(float)
for s, e, n, _ in seg: per[n] += e - s
print("streams=$S: 10 steps wall %.3f ms/step, device busy %.3f, idle %.3f, >=2 kernels in flight %.3f" % (wall / 1e7, busy / 1e7, (wall - busy) / 1e7, over / 1e7))
print("  per-kernel wall sums per step (ms):", {k: round(v / 1e7, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1])})
print("  queues:", collections.Counter(r[3] for r in seg))
PY
