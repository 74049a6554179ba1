#!/usr/bin/env python3
"""Turn the rocprofv3 (ROCm 7.2 writes rocpd SQLite) outputs that tools/gpu_profile.sh left under
gpurun_out/prof_<tag>/ into small tracked summaries under profiles/:
  profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average duration (== `rocprofv3 --kernel-trace --stats`)
  profiles/<tag>_pmc.csv            FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), KB per dispatch
  profiles/pmc_latest.json          HBM bytes per launch per kernel, raw and with the gfx950 FETCH_SIZE x2 correction
"""
import json
import os
import sqlite3
import sys

import re


def base(name):
    """'void k_fast<0>(FastParams)' -> 'k_fast': the instances of a template kernel (k_fast's two passes run once per batch each) are summed per batch"""
    b = re.sub(r"^void\s+", "", name).split("(")[0]
    return re.sub(r"<.*>$", "", b) if b.startswith("k_") else name.split("(")[0]   # (this library's kernels only)


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

c = sqlite3.connect(os.path.join(src, "stats", "trace_results.db"))
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only --streams 1  (durations in us)\n")
    f.write("kernel,calls,total_us,average_us,percentage\n")
    inst = {}
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        f.write("%s,%d,%.1f,%.3f,%.2f\n" % (r[0].replace(",", ";"), r[1], r[2], r[3], r[4]))
        inst.setdefault(base(r[0]), []).append(r)
    for b, rs in inst.items():
        if len(rs) > 1:   # a template kernel's instances, once per batch each: the stage is their sum
            calls = max(r[1] for r in rs)
            f.write("%s (sum of %d instances per batch),%d,%.1f,%.3f,%.2f\n" % (b, len(rs), calls, sum(r[2] for r in rs), sum(r[2] for r in rs) / calls, sum(r[4] for r in rs)))
    f.write("# launch geometry / registers (first dispatch of each kernel)\n")
    f.write("kernel,grid,workgroup,lds_bytes,vgpr,sgpr\n")
    for r in c.execute("select name,grid_x,grid_y,grid_z,workgroup_x,lds_size,vgpr_count,sgpr_count from kernels group by name"):
        f.write("%s,%dx%dx%d,%d,%d,%d,%d\n" % (r[0].replace(",", ";"), r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
pmc = {}
with open(os.path.join(dst, tag + "_pmc.csv"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), same command; values are KB per dispatch as reported\n")
    f.write("kernel,counter,dispatches,mean_KB_per_dispatch\n")
    for kind in ("fetch", "write"):
        d = sqlite3.connect(os.path.join(src, kind, "pmc_results.db"))
        rows_ = list(d.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"))
        # a stage = the instances of one template kernel; per batch it is every instance's mean x its launches per batch (k_fast<0> + k_fast<1>: one each;
        # k_resize2<true> once + k_resize2<false> six times) — launches per batch = dispatches / the fewest dispatches of any instance of the stage
        fewest = {}
        for name, ctr, n, mean in rows_:
            fewest[base(name)] = min(fewest.get(base(name), n), n)
        for name, ctr, n, mean in rows_:
            f.write("%s,%s,%d,%.1f\n" % (name.replace(",", ";"), ctr, n, mean))
            per_batch = n / float(fewest[base(name)]) if base(name) != name.split("(")[0] or "<" in name else 1.0
            pmc.setdefault(base(name), {})[ctr] = pmc.get(base(name), {}).get(ctr, 0.0) + mean * 1024.0 * per_batch
out = {"tag": tag, "note": "bytes per launch; FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads "
       "(MI355X_MICROARCH.md §HBM): 'corrected' doubles the read side; other widths are uncalibrated, so raw <= true <= corrected",
       "kernels": {}}
for k, v in pmc.items():
    fr, wr = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out["kernels"][k] = {"fetch_raw": fr, "write": wr, "traffic_raw": fr + wr, "traffic_corrected": 2 * fr + wr}
# which build / workload these passes belong to (bench.py refuses to quote `traffic` from a PMC pass of other kernel sources)
sys.path.insert(0, root)
import bench  # noqa: E402
out["source_sha"] = bench.source_sha()
out["workload"] = [752, 480, 1000, 512]

# SQ passes (tools/gpu_profile.sh sq1 / sq2): per-kernel means per dispatch + the derived figures the roofline discussion uses
sq = {}
for kind in ("sq1", "sq2"):
    fdb = os.path.join(src, kind, "pmc_results.db")
    if not os.path.exists(fdb):
        continue
    d = sqlite3.connect(fdb)
    for name, ctr, n, mean in d.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"):
        if base(name).startswith("k_"):
            sq.setdefault(base(name), {})[ctr] = sq.get(base(name), {}).get(ctr, 0.0) + mean
if sq:
    dur = {}
    for r in c.execute("select name,average from top_kernels"):
        dur[base(r[0])] = dur.get(base(r[0]), 0.0) + r[1]
    cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY",
            "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
    with open(os.path.join(dst, tag + "_sq.csv"), "w") as f:
        f.write("# rocprofv3 --pmc <8 SQ counters> (two separate passes, tools/gpu_profile.sh), same command; means per dispatch.  SQ_*_CYCLES / SQ_WAIT_* / "
                "SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md).  Derived: valu_per_wave = SQ_INSTS_VALU / SQ_WAVES; "
                "wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls); wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt / barrier); "
                "valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                "waves_per_simd = SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = mean resident waves per SIMD while the kernel runs (GRBM_GUI_ACTIVE is summed over the 8 XCDs)\n")
        f.write("kernel,avg_us," + ",".join(cols) + ",valu_per_wave,wait_inst_frac,wait_any_frac,valu_active_frac,lds_conflict_frac,waves_per_simd\n")
        for k in sorted(sq):
            v = sq[k]
            g = lambda n: v.get(n, float("nan"))
            wc = g("SQ_WAVE_CYCLES")
            derived = [g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc,
                       g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan"),
                       wc * 4.0 / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0) if g("GRBM_GUI_ACTIVE") else float("nan")]
            f.write("%s,%.1f,%s,%s\n" % (k, dur.get(k, float("nan")), ",".join("%.6g" % g(n) for n in cols), ",".join("%.4f" % x for x in derived)))
            # the same figures next to the HBM bytes, for bench.py's roofline object: resident waves per SIMD and the SIMD cycles one wave-level
            # VALU instruction costs (duration x 1024 SIMDs x 2.4 GHz / SQ_INSTS_VALU; 4 = a SIMD issuing one wave64 VALU instruction back to back)
            if k in out["kernels"] and k in dur and g("SQ_INSTS_VALU") == g("SQ_INSTS_VALU") and g("SQ_INSTS_VALU") > 0:
                out["kernels"][k].update({"valu_insts_per_wave": round(derived[0], 1), "waves_per_simd": round(derived[5], 2),
                                          "simd_cycles_per_valu_inst": round(dur[k] * 1e-6 * 1024 * 2.4e9 / g("SQ_INSTS_VALU"), 2),
                                          "lds_conflict_frac": round(derived[4], 3)})
    print(open(os.path.join(dst, tag + "_sq.csv")).read())
json.dump(out, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    import shutil
    shutil.copy(bj, os.path.join(dst, tag + "_bench.json"))
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read())
print(open(os.path.join(dst, tag + "_pmc.csv")).read())
