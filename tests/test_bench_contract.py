"""bench.py's output contract, checked without a GPU: the latest committed bench line (profiles/*_bench.json) carries every field the driver
and the judge read, names BASELINE.json's metric, and its roofline arithmetic is self-consistent with bench.algorithmic_bytes()."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    with open(files[-1]) as f:
        return json.load(f), files[-1]


def test_bench_line_fields():
    d, path = _latest()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == base["metric"], path
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, (k, path)
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # whole-job throughput and step time agree: frames per step / seconds per step
    frames_per_step = d["config"]["frames_per_gpu_per_step"] * d["n_gpus"]
    assert abs(d["value"] - frames_per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02


def test_headline_is_the_metric_it_names():
    """Round-1 verdict: `value` must be extract+match (the metric BASELINE.json names), timed over the full step count."""
    d, path = _latest()
    assert "extract+match" in d["value_is"] and "--steps" in d["value_is"], path
    mc = d["metric_components"]
    assert abs(mc["orb_extract_match_frames_per_s"] - d["value"]) < 1e-6 and mc["orb_extract_frames_per_s"] >= d["value"]
    assert mc["local_ba_linearizations_per_s"] > 0 and mc["local_ba_lm_iterations_per_s"] > 0
    assert "match" in d["config"]["workload"].lower() and "752x480" in d["config"]["workload"]
    r = d["roofline"]
    assert r["whole_step_algorithmic_bytes_per_frame"] == 3406774 + 184000            # A_ext + A_match
    assert abs(r["whole_step_frac"] - r["whole_step_algorithmic_bytes_per_frame"] * d["value"] / d["n_gpus"] / 1e9 / r["peak"]) < 1e-4
    assert set(r["per_kernel_frac"]) >= {"k_fast", "k_describe2", "k_resize2", "k_octree"}
    assert {"k_sbp_frame", "k_undistort_grid"} <= set(r["per_kernel_frac"]) or {"k_sbp_candidates2", "k_sbp_resolve"} <= set(r["per_kernel_frac"])   # round 4 / rounds 1-3
    c = d["cpu_baseline"]
    assert "SearchByProjection" in c["sample"] and c["per_core"] > 0 and c["cores"] <= (c.get("cpu_quota") or 1e9) * 2 + 1


def test_round4_fields_of_the_line():
    """median-of-regions headline with its spread, the oracle check of the timed step, the 1280x720 leg and the batch sweep, the measured VALU roof"""
    d, path = _latest()
    assert d["value_min"] <= d["value"] <= d["value_max"] and d["repeats"] >= 5 and len(d["region_ms"]) == d["repeats"], path
    assert abs(d["value"] - d["config"]["frames_per_gpu_per_step"] * d["n_gpus"] * d["steps"] / (sorted(d["region_ms"])[len(d["region_ms"]) // 2] * 1e-3)) / d["value"] < 1e-3
    c = d["config"]
    assert c["parity_checked_frames"] == c["frames_per_gpu_per_step"] * d["n_gpus"] and c["parity_mismatches"] == 0
    assert "bit-exact" in c["workload"]                      # only claimed next to a non-zero parity_checked_frames
    s = d["extra"]["size_1280x720"]
    assert s["nfeatures"] == 1500 and s["extract_match_frames_per_s"] > 0 and s["extract_frames_per_s"] >= s["extract_match_frames_per_s"] * 0.98
    assert s["whole_step_algorithmic_bytes_per_frame"] == 7084076 + 276000 and 0 < s["whole_step_frac"] < 1 and s["dominant_kernel"].startswith("k_")
    assert s["parity"]["mismatches"] == 0 and s["parity"]["checked_frames"] >= 32
    bs = d["extra"]["batch_sweep"]
    assert {"64", "512", "4096"} <= set(bs) and all(v["frames_per_s"] > 0 for v in bs.values())
    assert bs["4096"]["input_bytes"] > 256 * 1024 * 1024 and bs["4096"]["parity"]["mismatches"] == 0     # beyond the Infinity Cache
    r = d["roofline"]
    assert r["traffic"] is not None and r["traffic"] >= r["traffic_raw"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    oi = r["occupancy_and_issue"]
    assert 0.3 < oi["valu_issue_frac_of_measured_peak"] <= 1.0 and 2.0 < oi["valu_floor_cycles_per_inst_static_mix"] < 4.3


def test_round5_fields_of_the_line():
    """Round-4 verdict items 1, 3, 4, 6: the rank bookkeeping of a run (n_gpus = ranks that ran), the headline on rotating input sets of distinct scenes with
    the round-4 input next to it, the mixed batch with every frame checked, SURVEY's bytes in every per-kernel fraction, the single-window LM time."""
    d, path = _latest()
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and len(d["per_rank"]["frames_per_s"]) == d["n_gpus"], path
    assert abs(d["per_rank"]["min"] - d["value"]) / d["value"] < 0.02
    c = d["config"]
    assert c["input_sets"] >= 2 and c["distinct_scenes_per_set"] == c["frames_per_gpu_per_step"] // 2
    sd = d["extra"]["scene_diversity"]
    assert sd["headline_distinct_scenes"] == c["input_sets"] * c["distinct_scenes_per_set"] and 0.9 < sd["ratio_headline_over_r04_input"] < 1.1
    assert sd["r04_input_parity"]["mismatches"] == 0
    mb = d["extra"]["mixed_batch"]
    assert mb["parity"]["checked_frames"] == c["frames_per_gpu_per_step"] and mb["parity"]["mismatches"] == 0 and mb["frames_per_s"] > 0
    assert set(mb["frames_by_kind"]) == {"textured", "sparse", "low_contrast", "flat", "noise"} and mb["mean_keypoints_by_kind"]["flat"] == 0.0
    assert mb["frames_by_kind"]["textured"] > mb["frames_by_kind"]["sparse"] > mb["frames_by_kind"]["low_contrast"] > mb["frames_by_kind"]["noise"] > 0
    r = d["roofline"]
    assert r["traffic_ratio"] is not None and abs(r["traffic_ratio"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 2e-3
    B = c["frames_per_gpu_per_step"]
    want = 1000 * (961 + 512 + 60) * B / (d["kernel_ms"]["describe"] * 1e-3) / 1e9 / r["peak"]        # N (961 + 512 + 60), not the staged 43 x 43
    assert abs(r["per_kernel_frac"]["k_describe2"] - want) < 2e-3 and r["describe_patch_bytes_per_frame"] == 1000 * (43 * 43 + 60)
    lm = d["extra"]["lba"]
    assert lm["lm_single_window_ms_per_optimize5"] <= 3.0 and lm["lm_trials_per_window"] == 5.0      # round-4 verdict item 4 (was 4.72)


def test_valu_rates_table_is_committed():
    import csv
    rows = [r for r in csv.reader(open(os.path.join(ROOT, "profiles", "valu_rates.csv"))) if r and not r[0].startswith("#")]
    head, rows = rows[0], rows[1:]
    assert head[:4] == ["instruction", "mode", "waves_per_simd", "simd_cycles_per_inst"]
    rate = {(r[0], r[1], int(r[2])): float(r[3]) for r in rows}
    for op in ("v_add_u32", "v_and_b32", "v_fma_f32"):       # full rate with four or more waves per SIMD
        assert 1.9 < rate[(op, "indep8", 8)] < 2.6, op
    for op in ("v_lerp_u8", "v_pk_min_u16", "v_dot4_u32_u8", "v_alignbyte_b32", "v_perm_b32", "v_mad_u32_u24", "v_fma_f64", "v_lshlrev_b32"):
        assert 3.9 < rate[(op, "indep8", 8)] < 4.5, op      # half rate
    assert rate[("ds_bpermute_b32", "indep8", 8)] > 2.5 * rate[("ds_read_b32", "indep8", 8)]


def test_roofline_uses_algorithmic_bytes():
    sys.path.insert(0, ROOT)
    import bench
    d, _ = _latest()
    r = d["roofline"]
    if "algorithmic_bytes_per_launch" in r and "kernel_ms" in r:
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    whole, per_kernel = bench.algorithmic_bytes(752, 480, 1000)
    assert 3.3e6 < whole < 3.5e6                   # 3.41 MB of algorithmic traffic per 752x480 frame
    assert set(per_kernel) == {"pyramid", "fast", "octree", "describe"}
    if "algorithmic_bytes_per_launch" in r:        # the dominant kernel's bytes x the frames of one launch
        assert r["algorithmic_bytes_per_launch"] == per_kernel[r["kernel"].replace("k_", "")] * d["config"]["frames_per_gpu_per_step"]


def test_bench_cli_parses_without_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_gpus_flag_resolution():
    """Round-4 verdict, item 1: `--gpus N` without a launcher starts N ranks, with a launcher it must agree with WORLD_SIZE, and it never silently
    runs one rank."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resolve_world(None, {}) == ("run", 1) and bench.resolve_world(1, {}) == ("run", 1)
    assert bench.resolve_world(8, {}, visible_devices=8) == ("launch", 8)
    assert bench.resolve_world(8, {"WORLD_SIZE": "8"}) == ("run", 8) and bench.resolve_world(None, {"WORLD_SIZE": "4"}) == ("run", 4)
    with pytest.raises(SystemExit):
        bench.resolve_world(8, {"WORLD_SIZE": "1"})
    with pytest.raises(SystemExit):
        bench.resolve_world(1, {"WORLD_SIZE": "2"})
    with pytest.raises(SystemExit):
        bench.resolve_world(8, {}, visible_devices=1)             # more GPUs asked for than the node has: refuse, do not measure one
    assert bench.resolve_world(2, {"ORBHIP_BENCH_ONE_DEVICE": "1"}, visible_devices=1) == ("launch", 2)    # the documented one-device dry run
    with pytest.raises(SystemExit):
        bench.resolve_world(0, {})


def test_gpus_2_self_launch_starts_two_ranks():
    """`python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) re-executes under torch.distributed.run: two ranks form a group and rank 0 prints
    n_gpus = 2.  --launch-check stops before anything touches a GPU (gloo), so the launch logic itself runs in the CPU tier."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and sorted(r[0] for r in d["ranks"]) == [0, 1] and sorted(r[1] for r in d["ranks"]) == [0, 1]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "contradicts WORLD_SIZE" in bad.stderr


def test_inputs_of_the_headline():
    """make_batch keeps round 4's frames for the same arguments (committed fixtures and the GPU-tier parity tests depend on them); the mixed batch holds
    the five scene kinds in the stated shares, every pair from its own seed."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    a = bench.make_batch(6, seed0=3, unique=2, w=160, h=120)
    b = bench.make_batch(6, seed0=3, unique=2, w=160, h=120, workers=1)
    assert a.shape == (6, 120, 160) and np.array_equal(a, b)
    assert np.array_equal(a[4], np.roll(a[0], (7, 13), (0, 1)))                     # third pair = first scene again, rolled
    assert not np.array_equal(a[0], a[2])
    f, kinds = bench.make_mixed_batch(64, seed0=1, w=160, h=120, workers=2)
    assert f.shape == (64, 120, 160) and len(kinds) == 64 and kinds[0::2] == kinds[1::2]
    cnt = {k: kinds.count(k) // 2 for k in set(kinds)}
    assert cnt["textured"] == 19 and cnt["sparse"] == 6 and cnt["low_contrast"] == 3 and cnt["flat"] == 2 and cnt["noise"] == 2
    flat = [i for i, k in enumerate(kinds) if k == "flat"][0]
    assert f[flat].min() == f[flat].max()
    whole, pk = bench.algorithmic_bytes(752, 480, 1000)
    assert pk["describe"] == 1000 * (961 + 512 + 60) and bench.describe_patch_bytes(1000) == 1000 * (43 * 43 + 60)      # SURVEY 8(d)'s figure for the fraction


def test_source_sha_ignores_comments_only(tmp_path, monkeypatch):
    """profiles/pmc_latest.json is tied to the kernel sources by bench.source_sha(): a comment / white-space edit keeps the tie, a code
    edit (or an edit inside a string literal) breaks it."""
    import json
    import bench
    d = tmp_path / "awesome-orb-slam3-3dvisioncraft-version_amd" / "csrc"
    d.mkdir(parents=True)
    base = 'int f(int a) { return a / 2; }  // halves\nconst char* s = "a // b";\n'
    for f in ("orbx_extractor.hip", "orbm_matcher.hip", "orbf_frame.hip"):
        (d / f).write_text(base)
    real_root = bench.ROOT
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h0 = bench.source_sha()
    (d / "orbm_matcher.hip").write_text('/* block\n comment */ int f(int a)\n{\n    return a / 2;   // other words\n}\nconst char* s = "a // b";\n')
    assert bench.source_sha() == h0
    (d / "orbm_matcher.hip").write_text(base.replace("a / 2", "a / 3"))
    assert bench.source_sha() != h0
    (d / "orbm_matcher.hip").write_text(base.replace('"a // b"', '"a // c"'))
    assert bench.source_sha() != h0
    monkeypatch.setattr(bench, "ROOT", real_root)
    pm = json.load(open(os.path.join(real_root, "profiles", "pmc_latest.json")))
    if pm["source_sha"] != bench.source_sha():   # a kernel under development: bench.py then reports roofline.traffic as null, which is the honest line
        import warnings
        warnings.warn("profiles/pmc_latest.json belongs to other kernel sources: re-run tools/gpu_profile.sh + tools/summarize_prof.py")
