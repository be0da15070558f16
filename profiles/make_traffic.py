#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh into the committed summaries.

    python profiles/make_traffic.py <tag> [--key C2:K5:B32]

Reads  gpurun_out/<tag>_stats/**/*kernel_stats.csv, gpurun_out/<tag>_pmc_{fetch,write}/**/*counter_collection.csv
Writes profiles/<tag>_kernel_stats.csv        (rocprofv3 --kernel-trace --stats, copied)
       profiles/<tag>_pmc_hbm.csv             (per kernel: launches, FETCH_SIZE, WRITE_SIZE averages, corrected bytes)
       profiles/pmc_traffic.json              ({workload key: {bench kernel name: HBM bytes per launch}})

Corrections applied to the counters (gfx950): both are reported in KiB; FETCH_SIZE tallies each
128-byte memory-side read request as 64 bytes, so it is doubled (cross-check in this very trace:
points_fwd reads one fp32 per pixel and nothing else, and its raw FETCH_SIZE is half of 4 B/px);
WRITE_SIZE is taken as is.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# rocprofv3 kernel symbol -> the name bench.py's event profiler uses
NAMES = (
    ("code_pattern_fwd_roll_kernel", "code_pattern_fwd"), ("code_pattern_bwd_roll_kernel", "code_pattern_bwd"),
    ("code_pattern_bwd_max_kernel", "code_pattern_bwd_max"),
    ("code_pattern_fwd_wide_kernel", "code_pattern_fwd"), ("code_pattern_fwd_kernel", "code_pattern_fwd"),
    ("code_pattern_norm_kernel", "code_pattern_norm"), ("code_pattern_bwd_reduce_kernel", "code_pattern_bwd_reduce"),
    ("code_pattern_bwd_wide_kernel", "code_pattern_bwd"), ("code_pattern_bwd_kernel", "code_pattern_bwd"),
    ("pose_transforms_kernel", "pose_transforms"),
    # layout 3 = points formed from the depth plane (reported apart: it moves 12 B/px less)
    ("project_kernel<3", "warp_project_splat_from_depth"), ("warp_sample_kernel<3", "warp_sample_from_depth"),
    ("project_kernel", "warp_project_splat"), ("warp_sample_kernel", "warp_sample"), ("bilinear_bwd_img_scatter_kernel", "bilinear_bwd_img_scatter"),
    ("bilinear_bwd_img_gather_kernel", "bilinear_bwd_img_gather"), ("bilinear_tap_index_kernel", "bilinear_tap_index"),
    ("project_bwd_transform_sum_kernel", "project_bwd_transform_sum"), ("project_bwd_transform_kernel", "project_bwd_transform"),
    ("points_fwd_kernel", "points_fwd"),
    ("code_pattern_fwd_sweep_kernel", "code_pattern_fwd"), ("code_pattern_bwd_sweep_kernel", "code_pattern_bwd"),
    ("code_pattern_flat_fwd_kernel", "code_pattern_fwd"), ("code_pattern_flat_bwd_kernel", "code_pattern_bwd"),
    ("zero_fill_kernel", "zero_fill"), ("reflectance_kernel", "reflectance"),
    ("capture_project_kernel", "capture_project_splat"), ("capture_sample_kernel", "capture_sample"),
    ("capture_peak_kernel", "capture_peak"), ("capture_sensor_kernel", "capture_sensor"),
)


def bench_name(sym):
    for frag, name in NAMES:
        if frag in sym:
            return name
    return None


def counters(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True),
               key=os.path.getmtime, reverse=True)        # the latest run under this tag
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    with open(f[0]) as fh:
        for r in csv.DictReader(fh):
            n = bench_name(r["Kernel_Name"])
            if n:
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--key", default="C2:K5:B32")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    src = os.path.join(ROOT, "gpurun_out")
    st = sorted(glob.glob(os.path.join(src, a.tag + "_stats", "**", "*kernel_stats.csv"), recursive=True),
                key=os.path.getmtime, reverse=True)
    if st:
        shutil.copy(st[0], os.path.join(out, a.tag + "_kernel_stats.csv"))
    fe = counters(os.path.join(src, a.tag + "_pmc_fetch"))
    wr = counters(os.path.join(src, a.tag + "_pmc_write"))
    rows, traffic = [], {}
    for n in sorted(set(fe) | set(wr)):
        f = fe[n].get("FETCH_SIZE", [])
        w = wr[n].get("WRITE_SIZE", [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        rd_b, wr_b = 2.0 * fa * 1024.0, wa * 1024.0
        rows.append(dict(kernel=n, launches=max(len(f), len(w)), FETCH_SIZE_KiB_avg=round(fa, 1), WRITE_SIZE_KiB_avg=round(wa, 1),
                         read_bytes_corrected=int(rd_b), write_bytes=int(wr_b), hbm_bytes_per_launch=int(rd_b + wr_b)))
        traffic[n] = int(rd_b + wr_b)
    with open(os.path.join(out, a.tag + "_pmc_hbm.csv"), "w", newline="") as fh:
        wtr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wtr.writeheader()
        wtr.writerows(rows)
    tp = os.path.join(out, "pmc_traffic.json")
    allt = {}
    if os.path.exists(tp):
        with open(tp) as fh:
            allt = json.load(fh)
    allt[a.key] = traffic
    allt.setdefault("_source", {})[a.key] = f"profiles/{a.tag}_pmc_hbm.csv"
    with open(tp, "w") as fh:
        json.dump(allt, fh, indent=1, sort_keys=True)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
