#!/bin/bash
# Runs ON THE GPU BOX: LDS-side counters of k_fast / k_describe for every prebuilt variant under exp_so/ (see tools/exp_pmc_variants.sh for the issue side)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for so in $R/exp_so/*.so; do
  n=$(basename $so .so)
  ORBHIP_LIB=$so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcl_$n -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only --streams 1 > /tmp/pmcl_$n.log 2>&1
  f=$(find /tmp/pmcl_$n -name "*counter_collection.csv" | head -1)
  python - "$f" "$n" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if k.startswith("k_fast") or k.startswith("k_describe"):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(sys.argv[2], k, {c: "%.4g" % x for c, x in m.items()}, "valu/wave %.1f  lds conflict frac %.3f  lds busy %.3f" % (
        m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_LDS_IDX_ACTIVE"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256)))
PY
done
