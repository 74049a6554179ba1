R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in $R/exp_so/*.so; do
  n=$(basename $so .so)
  echo "== $n"
  ORBHIP_LIB=$so timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -o t -- python $R/tools/exp.py lm --reps 1 $LMARGS > /tmp/log_$n.txt 2>&1
  grep lm_iter /tmp/log_$n.txt
  f=$(find /tmp/p_$n -name '*kernel_stats.csv' | head -1)
  python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %9.1f min %9.1f max %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
