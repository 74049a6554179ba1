R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for W in 8 32 256; do
  rm -rf /tmp/lf_$W
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/lf_$W -- python $R/tools/exp_lm.py --reps 1 --windows $W > /tmp/lf_$W.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ls_$W -- python $R/tools/exp_lm.py --reps 1 --windows $W > /tmp/ls_$W.log 2>&1
  python - $W <<'PY'
import csv,glob,sys
W=sys.argv[1]
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/lf_%s/**/*counter_collection.csv"%W,recursive=True) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_lm_schur","void k_lm_schur"))]
d=[float(r["AverageNs"]) for f in glob.glob("/tmp/ls_%s/**/*kernel_stats.csv"%W,recursive=True) for r in csv.DictReader(open(f)) if "k_lm_schur" in r["Name"]]
print("windows",W,"schur FETCH_SIZE raw MB/launch %.0f  per window %.1f MB ; avg us %s"%(sum(v)/len(v)/1024, sum(v)/len(v)/1024/int(W), d))
PY
done
