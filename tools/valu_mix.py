#!/usr/bin/env python3
"""Static VALU instruction mix of the product kernels against the measured issue rates (profiles/valu_rates.csv, tools/ubench/valu_rate.hip):
which share of a kernel's vector instructions are full-rate forms (2.15 SIMD cycles per wave64 instruction: add / sub / and / or / xor / not / mov /
right shifts / f32 add-mul-fma), half-rate (4.2: everything else 32-bit, packed 16-bit, dot, DPP/SDWA, 64-bit) or quarter-rate (8.1: rcp, sqrt,
sin, ...), and the issue-cycle floor per instruction that mix implies.  Two counts per kernel: STATIC (every instruction once) and LOOP-WEIGHTED
(an instruction inside a loop of depth d counts LOOP_WEIGHT^d times — LLVM annotates every basic block of the .s with its loop and depth; fully
unrolled bodies carry no depth and count once).  The static mix is dominated by prologues and address set-up (add / mov: full rate), the hot
loops are byte / packed / dot forms (half rate): the weighted floor is the better estimate of what the SIMDs execute, the static one a lower
bound of it.  Neither is a dynamic profile (no per-block execution counts exist on this stack: the PMC type counters put every integer form in
one class).   usage: tools/valu_mix.py <file.s from hipcc -S --cuda-device-only> ... [kernel substring ...] [--json profiles/valu_mix.json]"""
import collections
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_u16", "v_sub_u16", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_add_nc_u32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
RATE = {"full": 2.15, "half": 4.2, "quarter": 8.1}
LOOP_WEIGHT = 8     # assumed trips per loop level (the kernels' loops run 4 .. 40 turns)


def main():
    args = sys.argv[1:]
    jout = None
    if "--json" in args:
        i = args.index("--json")
        jout = args[i + 1]
        del args[i:i + 2]
    paths = [a for a in args if a.endswith(".s")]
    want = [a for a in args if not a.endswith(".s")]
    result = {}
    for path in paths:
        one(path, want, result)
    if jout:
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        json.dump({"source_sha": bench.source_sha(), "rates_simd_cycles_per_wave64_instruction": RATE,
                   "note": "static instruction mix of the compiled kernels (tools/valu_mix.py on hipcc -S output) priced with the measured issue rates of "
                           "profiles/valu_rates.csv: floor = the SIMD cycles per VALU instruction the mix allows when nothing but issue limits the kernel",
                   "kernels": result}, open(jout, "w"), indent=1)


def demangle(k):
    m = re.match(r"^_ZL?(\d+)(k_\w+)", k)
    if not m:
        return k
    name = m.group(2)[:int(m.group(1))]
    t = re.search(r"ILi(\d+)E", k) or re.search(r"ILb(\d)E", k)
    return name + ("<%s>" % t.group(1) if t else "")


def classify(valu):
    base = lambda o: re.sub(r"_(e32|e64|dpp|sdwa)$", "", o)
    cls = collections.Counter()
    for o, n in valu.items():
        b = base(o)
        variant = o.endswith("_dpp") or o.endswith("_sdwa")
        cls["quarter" if b in QUARTER else ("full" if b in FULL and not variant else "half")] += n
    tot = sum(valu.values())
    return cls, (sum(RATE[x] * n for x, n in cls.items()) / tot if tot else 0.0)


def one(path, want, result):
    cur, mix, wmix, depth, after_label = None, collections.OrderedDict(), {}, 0, False
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            mix[cur] = collections.Counter(); wmix[cur] = collections.Counter()
            depth, after_label = 0, False
            continue
        if cur is None:
            continue
        if ".end_amdhsa_kernel" in line or line.startswith("\t.section"):
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):          # a basic block: its loop depth follows in the label's comment lines
            depth, after_label = 0, True
            d = re.search(r"Depth=(\d+)", line)
            if d:
                depth = int(d.group(1))
            continue
        if after_label and re.match(r"^\s*;", line):
            d = re.search(r"Depth=(\d+)", line)
            if d:
                depth = max(depth, int(d.group(1)))
            continue
        m = re.match(r"^\s+(v_\w+|ds_\w+|s_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)", line)
        if m:
            after_label = False
            mix[cur][m.group(1)] += 1
            wmix[cur][m.group(1)] += LOOP_WEIGHT ** depth
    for k, c in mix.items():
        if want and not any(w in k for w in want):
            continue
        valu = {o: n for o, n in c.items() if o.startswith("v_")}
        tot = sum(valu.values())
        if tot == 0:
            continue
        cls, floor = classify(valu)
        wvalu = {o: n for o, n in wmix[k].items() if o.startswith("v_")}
        wcls, wfloor = classify(wvalu)
        wtot = sum(wvalu.values())
        result[demangle(k)] = {"valu": tot, "full_rate": cls["full"], "half_rate": cls["half"], "quarter_rate": cls["quarter"], "floor_cycles_per_inst": round(floor, 3),
                               "loop_weighted": {"loop_weight": LOOP_WEIGHT, "full_rate_share": round(wcls["full"] / wtot, 4), "half_rate_share": round(wcls["half"] / wtot, 4),
                                                 "quarter_rate_share": round(wcls["quarter"] / wtot, 4), "floor_cycles_per_inst": round(wfloor, 3)}}
        print("%s: %d VALU (full-rate %.1f %%, half %.1f %%, quarter %.1f %%) -> %.2f SIMD cycles per instruction at the measured rates; loop-weighted (x%d per level): "
              "full-rate %.1f %% -> %.2f; %d LDS, %d SALU, %d VMEM"
              % (k, tot, 100.0 * cls["full"] / tot, 100.0 * cls["half"] / tot, 100.0 * cls["quarter"] / tot, floor, LOOP_WEIGHT, 100.0 * wcls["full"] / wtot, wfloor,
                 sum(n for o, n in c.items() if o.startswith("ds_")), sum(n for o, n in c.items() if o.startswith("s_")),
                 sum(n for o, n in c.items() if o.startswith(("global_", "buffer_", "flat_", "scratch_")))))
        print("   top:", ", ".join("%s %d" % (o, n) for o, n in sorted(valu.items(), key=lambda kv: -kv[1])[:14]))


if __name__ == "__main__":
    main()
