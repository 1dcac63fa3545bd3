#!/usr/bin/env python3
"""Developer tool: turn the output of tools/collect_profiles.sh (gpurun_out/<tag>/) into the committed files under
profiles/ (prefix r02_): copies the summaries, reduces the PMC CSVs to the kernels of this repo, and derives
profiles/r02_pairwise_traffic.json (HBM bytes per launch of the pairwise kernels, with the SHA-256 of scp_kernels.hip the
counters were collected on -- bench.py reports roofline.traffic only while that still matches).

    python tools/postprocess_profiles.py r02c
"""
import csv
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
PFX = os.environ.get("PFX", "r03_")

COPIES = {
    "bench_n1024.json": "bench_n1024.json", "bench_n1024_kernel_stats.csv": "bench_n1024_kernel_stats.csv",
    "bench_n4096_1gpu.json": "bench_n4096_1gpu.json", "phase_profile_n1024.txt": "phase_profile_n1024.txt",
    "full_solve_timing.txt": "full_solve_timing.txt", "ref_config_timing.txt": "ref_config_timing.txt",
    "soak_80.txt": "soak_80_random_solves.txt", "soak_80_polish.txt": "soak_80_random_solves_polish.txt",
    "soak_24_large.txt": "soak_24_large_solves.txt",
    "batch128_rates.txt": "batch128_rates.txt", "batch128_cold.txt": "batch128_cold.txt", "demo_k500.txt": "demo_k500.txt",
    "grid_sync_bench.txt": "grid_sync_bench.txt", "solve128.txt": "solve128.txt",
    "solve128_kernel_stats.csv": "solve128_kernel_stats.csv",
    "batch128_with_events.txt": "batch128_with_events.txt", "batch128_multi_launch_passes.txt": "batch128_multi_launch_passes.txt",
    "batch128_records_4x5.txt": "batch128_records_4x5.txt", "solve128_timeline.txt": "solve128_timeline.txt",
    "small_pass_profile_n128.txt": "small_pass_profile_n128.txt", "small_pass_profile_n256.txt": "small_pass_profile_n256.txt",
    "bench_n4096_1gpu_kernel_stats.csv": "bench_n4096_1gpu_kernel_stats.csv",
    "phase_profile_lean_n4096.txt": "phase_profile_lean_n4096.txt", "step_time.txt": "step_time.txt",
    "phase_profile_round2_kernel_n1024.txt": "phase_profile_round2_kernel_n1024.txt",
    "bench_n1024_under_rocprof.json": "bench_n1024_under_rocprof.json",
    "rehearsal_2ranks_gloo_n1024.json": "rehearsal_2ranks_gloo_n1024.json",
    "rehearsal_2ranks_gloo_n4096.json": "rehearsal_2ranks_gloo_n4096.json",
    "pair_context_clock_1024.txt": "pair_context_clock_1024.txt", "membw.txt": "membw.txt",
    "pair_timeline_equal_chunks.txt": "pair_timeline_equal_chunks.txt", "pair_timeline_items.txt": "pair_timeline_items.txt",
    "pair_timeline_items_nostores.txt": "pair_timeline_items_nostores.txt",
    "pair_bench_n4096.txt": "pair_bench_n4096.txt", "pair_bench_n1024.txt": "pair_bench_n1024.txt",
}
for a, b in COPIES.items():
    pa = os.path.join(src, a)
    if os.path.exists(pa) and os.path.getsize(pa) > 0:
        shutil.copyfile(pa, os.path.join(dst, PFX + b))
    else:
        print("missing:", a)

OURS = ("pair_pass_kernel", "cg1_", "add_rows_at", "qp0_col", "compact_", "csr_", "pair_prep", "qp_reset", "kinematics", "rows_value",
        "add_rows", "reset_install", "spd_inverse", "gemm_f64", "pack_operands", "build_hf", "rel_step", "check_done")


def reduce_pmc(name_in, name_out):
    """keep only this repo's kernels, drop per-dispatch ids: kernel, grid, workgroup, counter, value"""
    pa = os.path.join(src, name_in)
    if not os.path.exists(pa):
        print("missing:", name_in)
        return []
    rows = []
    with open(pa) as f:
        for r in csv.DictReader(f):
            if any(k in r["Kernel_Name"] for k in OURS):
                rows.append(r)
    with open(os.path.join(dst, PFX + name_out), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count",
                    "Counter_Name", "Counter_Value", "Duration_ns"])
        for r in rows:
            w.writerow([r["Kernel_Name"][:120], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                        r["Accum_VGPR_Count"], r["Counter_Name"], r["Counter_Value"],
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    return rows


w_rows = reduce_pmc("pairwise_pmc_WRITE_SIZE.csv", "pairwise_pmc_WRITE_SIZE.csv")
f_rows = reduce_pmc("pairwise_pmc_FETCH_SIZE.csv", "pairwise_pmc_FETCH_SIZE.csv")
reduce_pmc("pairwise_pmc_TCC.csv", "pairwise_pmc_TCC.csv")
reduce_pmc("qp_pmc_mfma_raw.csv", "qp_mfma_counters.csv")
reduce_pmc("qp_pmc_lds_raw.csv", "lds_bank_conflicts.csv")


def per_launch(rows, counter, needle):
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    v = v[-3:]  # the timed repetitions of tools/pair_bench.py (the first launches include warm-up sizes)
    return (sum(v) / len(v) * 1024.0, len(v)) if v else (None, 0)  # the counters are in KB


N, K, D = 1024, 50, 2
rows_n = N * (N - 1) // 2 * K
kinds = {"linearize": ("pair_pass_kernel<2, 0,", rows_n * 8 * (D + 1) + 2 * N * K * D * 8),
         "violations_recompute": ("pair_pass_kernel<2, 3,", 2 * N * K * D * 8),
         "check": ("pair_pass_kernel<2, 1,", 2 * N * K * D * 8),
         "select": ("pair_pass_kernel<2, 4,", N * K * D * 8)}
out = {
    "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, --kernel-trace only) -- python3 "
              "tools/pair_bench.py --reps 3; 1024 agents x 50 steps, D=2 (tools/collect_profiles.sh, "
              "tools/postprocess_profiles.py)",
    "corrections": "units are KB; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read stream, "
                   "MI355X_MICROARCH.md HBM section); WRITE_SIZE exact for 16-B-per-lane stores",
    "kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, "ba-path-planning_amd", "csrc", "scp_kernels.hip"),
                                                "rb").read()).hexdigest(),
}
ok = True
for key, (needle, alg) in kinds.items():
    wb, nw = per_launch(w_rows, "WRITE_SIZE", needle)
    fb, nf = per_launch(f_rows, "FETCH_SIZE", needle)
    if wb is None or fb is None:
        print("no counter rows for", key)
        ok = False
        continue
    out[key] = {"rows": rows_n, "algorithmic_bytes": alg, "write_bytes": wb, "fetch_bytes": 2.0 * fb,
                "hbm_bytes": wb + 2.0 * fb, "launches_averaged": min(nw, nf)}
# config 4 (4096 x 50, 2-D: the L1 / L2 path of the same kernels) from its own pair of passes, when they were collected
w4 = reduce_pmc("pairwise_pmc_WRITE_SIZE_n4096.csv", "pairwise_pmc_WRITE_SIZE_n4096.csv")
f4 = reduce_pmc("pairwise_pmc_FETCH_SIZE_n4096.csv", "pairwise_pmc_FETCH_SIZE_n4096.csv")
if w4 and f4:
    N4 = 4096
    rows4 = N4 * (N4 - 1) // 2 * K
    cfg = {}
    for key, (needle, _) in kinds.items():
        wb, nw = per_launch(w4, "WRITE_SIZE", needle)
        fb, nf = per_launch(f4, "FETCH_SIZE", needle)
        if wb is None or fb is None:
            continue
        alg4 = {"linearize": rows4 * 8 * (D + 1) + 2 * N4 * K * D * 8, "select": N4 * K * D * 8}.get(key, 2 * N4 * K * D * 8)
        cfg[key] = {"rows": rows4, "algorithmic_bytes": alg4, "write_bytes": wb, "fetch_bytes": 2.0 * fb,
                    "hbm_bytes": wb + 2.0 * fb, "launches_averaged": min(nw, nf)}
    if "linearize" in cfg:
        out["configs"] = {"4096x50x2": cfg}
if ok:
    with open(os.path.join(dst, PFX + "pairwise_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["hbm_bytes"] for k, v in out.items() if isinstance(v, dict) and "hbm_bytes" in v}))
