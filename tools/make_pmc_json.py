#!/usr/bin/env python
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and the --stats pass of tools/profile_round.sh into
profiles/r02_k_inner_pmc.json + profiles/r02_bench_kernel_stats.csv.  Run where gpurun_out/ has been merged back.
    python tools/make_pmc_json.py [commit]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def per_launch(pattern, counter, kernel="k_inner<128, 512, 0>"):
    files = glob.glob(os.path.join(OUT, pattern, "**", "*counter_collection.csv"), recursive=True)
    vals = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.setdefault(row["Dispatch_Id"], 0.0)
                    vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    v = sorted(vals.values())
    return (sum(v) / len(v) if v else None), len(v)


def main():
    commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
    fetch, nf = per_launch("prof_r02_fetch", "FETCH_SIZE")
    write, nw = per_launch("prof_r02_write", "WRITE_SIZE")
    stats = glob.glob(os.path.join(OUT, "prof_r02", "**", "*kernel_stats.csv"), recursive=True)
    avg_us = None
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", "r02_bench_kernel_stats.csv"))
        with open(stats[0]) as fh:
            for row in csv.DictReader(fh):
                if "k_inner<128, 512, 0>" in row["Name"]:
                    avg_us = float(row["AverageNs"]) / 1e3
    out = {"kernel": "k_inner<128,512,0>", "workload": "C2: 256 tasks, N=Nq=128, d=256, I=20", "commit": commit,
           "avg_duration_us": avg_us, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "pmc_launches": [nf, nw],
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python bench.py --steps 5 --warmup 2`; KB as reported, summed over "
                   "XCDs by rocprofv3; bench.py applies the guide's gfx950 correction (2 x FETCH_SIZE: the D2 reads are 16-byte-per-lane streams); "
                   "compulsory traffic of the launch: read D2 (16.8 MB) + write A^-1 (16.8 MB) + vectors"}
    with open(os.path.join(ROOT, "profiles", "r02_k_inner_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out)


if __name__ == "__main__":
    main()
