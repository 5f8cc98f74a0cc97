#!/usr/bin/env python
"""Turns the rocprofv3 passes of tools/round3_profiles.sh (merged back under gpurun_out/) into the tracked evidence:
profiles/r03_bench_kernel_stats.csv, r03_k_inner_pmc.json (FETCH_SIZE / WRITE_SIZE per launch), r03_k_inner_sq_pmc.json
(SQ, LDS and MFMA counters per launch), r03_c5_kernel_stats.csv and r03_c5_fit_pmc.json (traffic of one blocked fit).
    python tools/make_pmc_json.py [commit]            (round 3 layout)
    python tools/make_pmc_json.py r04 [commit]        (round 4: tools/round4_profiles.sh; k_inner AND k_hyper, profiles/r04_*)
    python tools/make_pmc_json.py c5 [commit]         (round 4: tools/r04_c5_pmc.sh; traffic of one blocked fit by dispatch order)
    python tools/make_pmc_json.py r05 <commit>        (round 5: tools/r05_round_end.sh, ON the GPU box: summaries into gpurun_out/r05_*; the
                                                       raw rocprofv3 directories are deleted there)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
KINNER = "k_inner<128, 512, 0"


def counter_rows(pattern):
    for f in glob.glob(os.path.join(OUT, pattern, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


def per_launch(pattern, counters, kernel=KINNER):
    """{counter: mean over the kernel's dispatches of the counter summed over its rows}, number of dispatches"""
    vals = {c: {} for c in counters}
    for row in counter_rows(pattern):
        if kernel in row["Kernel_Name"] and row["Counter_Name"] in vals:
            d = vals[row["Counter_Name"]]
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    out = {c: (sum(d.values()) / len(d) if d else None) for c, d in vals.items()}
    return out, max((len(d) for d in vals.values()), default=0)


def stats_avg_us(stats_csv, name):
    with open(stats_csv) as fh:
        for row in csv.DictReader(fh):
            if name in row["Name"]:
                return float(row["AverageNs"]) / 1e3, int(row["Calls"])
    return None, 0


def main_r04(commit):
    prof = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(OUT, "prof_r04", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(prof, "r04_bench_kernel_stats.csv"))
    stats5 = glob.glob(os.path.join(OUT, "prof_r04_c5", "**", "*kernel_stats.csv"), recursive=True)
    if stats5:
        shutil.copy(stats5[0], os.path.join(prof, "r04_c5_kernel_stats.csv"))
    sq_names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]
    lds_names = ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
    for tag, kernel, what in (("k_inner", KINNER, "read D2 (16.8 MB) + write A^-1 (16.8 MB) + vectors"),
                              ("k_hyper", "k_hyper<", "read A^-1 twice, D2_ss / D2_qs / D2_qq (several passes, L2 hits after the first), the parked "
                                                      "exponential factors; write W_ss, W_qs, W_qq and the parked factors: 7 x 16.8 MB compulsory")):
        avg_us = stats_avg_us(stats[0], kernel)[0] if stats else None
        fetch, nf = per_launch("prof_r04_fetch", ["FETCH_SIZE"], kernel)
        write, nw = per_launch("prof_r04_write", ["WRITE_SIZE"], kernel)
        out = {"kernel": tag, "workload": "C2: 256 tasks, N=Nq=128, d=256, I=20", "commit": commit, "avg_duration_us": avg_us,
               "FETCH_SIZE_KB_per_launch": fetch["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": write["WRITE_SIZE"], "pmc_launches": [nf, nw],
               "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters alone with --kernel-trace) of `python bench.py --steps 5 "
                       "--warmup 2`; KB as reported, summed over XCDs by rocprofv3; bench.py applies the guide's gfx950 correction (2 x FETCH_SIZE for "
                       "16-byte-per-lane streaming reads); compulsory traffic of the launch: " + what}
        with open(os.path.join(prof, f"r04_{tag}_pmc.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(out)
        sq, n1 = per_launch("prof_r04_sq", sq_names, kernel)
        lds, n2 = per_launch("prof_r04_lds", lds_names, kernel)
        sqo = {"kernel": tag, "workload": out["workload"], "commit": commit, "per_launch": {**sq, **{k: v for k, v in lds.items() if k not in sq}},
               "launches": [n1, n2],
               "note": "two rocprofv3 --pmc passes (7 SQ/GRBM counters each); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over "
                       "waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs (MI355X_MICROARCH.md)"}
        if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAIT_ANY") is not None:
            sqo["wait_any_frac_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
            sqo["valu_issue_frac_of_wave_cycles"] = (sq.get("SQ_ACTIVE_INST_VALU") or 0.0) / sq["SQ_WAVE_CYCLES"]
        if lds.get("SQ_VALU_MFMA_BUSY_CYCLES") and sq.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs run for GRBM / 8 cycles each
            sqo["mfma_busy_frac_of_simd_cycles"] = lds["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        with open(os.path.join(prof, f"r04_{tag}_sq_pmc.json"), "w") as fh:
            json.dump(sqo, fh, indent=1)
        print(sqo)


def main_c5(commit):
    """Round 4, configuration 5: traffic of ONE blocked inner fit, attributed by dispatch order - the dispatches from k_lg_begin up to
    the last k_lg_advance before a kernel that is not part of the fit (the blocked sweep of Sigma_q in the outer stage uses the same
    k_lg_diag / ProbLg* kernels and is NOT counted).  Input: tools/r04_c5_pmc.sh."""
    prof = os.path.join(ROOT, "profiles")
    stats5 = glob.glob(os.path.join(OUT, "prof_r04_c5b", "**", "*kernel_stats.csv"), recursive=True)
    if stats5:
        shutil.copy(stats5[0], os.path.join(prof, "r04_c5_kernel_stats.csv"))
    fit_set = ("k_lg_begin", "k_lg_build", "k_lg_diag", "ProbLgPanel", "ProbLgUpdate", "k_lg_matvec", "k_lg_traces", "k_lg_advance")

    def fit_totals(pattern, counter):
        per = {}
        for row in counter_rows(pattern):
            if row["Counter_Name"] == counter:
                d = per.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])
                d[1] += float(row["Counter_Value"])
        fits, cur, in_fit, launches = [], 0.0, False, 0
        for did in sorted(per):
            name, val = per[did]
            is_fit = any(k in name for k in fit_set)
            if "k_lg_begin" in name:
                in_fit, cur, launches = True, 0.0, 0
            if in_fit and not is_fit:
                fits.append((cur, launches))
                in_fit = False
            if in_fit:
                cur += val
                launches += 1
        return fits

    f, w = fit_totals("prof_r04_c5_fetch", "FETCH_SIZE"), fit_totals("prof_r04_c5_write", "WRITE_SIZE")
    c5 = {"workload": "C5: 8 tasks, N=Nq=1024, d=512, I=20 (blocked path, csrc/large.h)", "commit": commit,
          "FETCH_SIZE_KB_per_fit": sum(v for v, _ in f) / len(f) if f else None, "WRITE_SIZE_KB_per_fit": sum(v for v, _ in w) / len(w) if w else None,
          "fits": [len(f), len(w)], "launches_per_fit": f[0][1] if f else None,
          "note": "dispatches of one adkf_fit call by dispatch order (k_lg_begin .. last k_lg_advance): 20 evaluations x (k_lg_build, 8 block steps of "
                  "k_lg_diag / ProbLgPanel / ProbLgUpdate, k_lg_matvec, k_lg_traces, k_lg_advance); the blocked sweep of Sigma_q in the outer stage is "
                  "not part of it; separate rocprofv3 --pmc passes; KB as reported by rocprofv3, bench.py applies 2 x FETCH_SIZE + WRITE_SIZE"}
    with open(os.path.join(prof, "r04_c5_fit_pmc.json"), "w") as fh:
        json.dump(c5, fh, indent=1)
    print(c5)


def main_r05(commit):
    """Round 5: like r04 + c5, from gpurun_out/prof_r05*; everything is written to gpurun_out/r05_* (copied to profiles/ afterwards)."""
    def first(pattern):
        g = glob.glob(os.path.join(OUT, pattern, "**", "*kernel_stats.csv"), recursive=True)
        return g[0] if g else None
    stats, stats5 = first("prof_r05"), first("prof_r05_c5")
    if stats:
        shutil.copy(stats, os.path.join(OUT, "r05_bench_kernel_stats.csv"))
    if stats5:
        shutil.copy(stats5, os.path.join(OUT, "r05_c5_kernel_stats.csv"))
    sq_names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]
    lds_names = ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
    for tag, kernel, what in (("k_inner", KINNER, "read D2 (16.8 MB) + write A^-1 (16.8 MB) + vectors"),
                              ("k_hyper", "k_hyper<", "read A^-1 twice, D2_ss / D2_qs / D2_qq (several passes, L2 hits after the first), the parked "
                                                      "exponential factors; write W_ss, W_qs, W_qq and the parked factors: 7 x 16.8 MB compulsory")):
        avg_us = stats_avg_us(stats, kernel)[0] if stats else None
        fetch, nf = per_launch("prof_r05_fetch", ["FETCH_SIZE"], kernel)
        write, nw = per_launch("prof_r05_write", ["WRITE_SIZE"], kernel)
        out = {"kernel": tag, "workload": "C2: 256 tasks, N=Nq=128, d=256, I=20", "commit": commit, "avg_duration_us": avg_us,
               "FETCH_SIZE_KB_per_launch": fetch["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": write["WRITE_SIZE"], "pmc_launches": [nf, nw],
               "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters alone with --kernel-trace) of `python bench.py --steps 5 "
                       "--warmup 2`; KB as reported, summed over XCDs by rocprofv3; bench.py applies the guide's gfx950 correction (2 x FETCH_SIZE for "
                       "16-byte-per-lane streaming reads); compulsory traffic of the launch: " + what}
        with open(os.path.join(OUT, f"r05_{tag}_pmc.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(out)
        sq, n1 = per_launch("prof_r05_sq", sq_names, kernel)
        lds, n2 = per_launch("prof_r05_lds", lds_names, kernel)
        sqo = {"kernel": tag, "workload": out["workload"], "commit": commit, "per_launch": {**sq, **{k: v for k, v in lds.items() if k not in sq}},
               "launches": [n1, n2],
               "note": "two rocprofv3 --pmc passes (7 SQ/GRBM counters each); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over "
                       "waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs (MI355X_MICROARCH.md)"}
        if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAIT_ANY") is not None:
            sqo["wait_any_frac_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
            sqo["valu_issue_frac_of_wave_cycles"] = (sq.get("SQ_ACTIVE_INST_VALU") or 0.0) / sq["SQ_WAVE_CYCLES"]
        if lds.get("SQ_VALU_MFMA_BUSY_CYCLES") and sq.get("GRBM_GUI_ACTIVE"):
            sqo["mfma_busy_frac_of_simd_cycles"] = lds["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        with open(os.path.join(OUT, f"r05_{tag}_sq_pmc.json"), "w") as fh:
            json.dump(sqo, fh, indent=1)
        print(sqo)
    # C5: traffic and launches of ONE blocked inner fit by dispatch order (the Sigma_q sweep of the outer stage is not part of it)
    fit_set = ("k_lg_begin", "k_lg_build", "k_lg_diag", "ProbLgPanel", "ProbLgUpdate", "k_lg_update_sweep", "k_lg_matvec", "k_lg_traces", "k_lg_advance")

    def fit_totals(pattern, counter):
        per = {}
        for row in counter_rows(pattern):
            if row["Counter_Name"] == counter:
                d = per.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])
                d[1] += float(row["Counter_Value"])
        fits, cur, in_fit, launches = [], 0.0, False, 0
        for did in sorted(per):
            name, val = per[did]
            is_fit = any(k in name for k in fit_set)
            if "k_lg_begin" in name:
                in_fit, cur, launches = True, 0.0, 0
            if in_fit and not is_fit:
                fits.append((cur, launches))
                in_fit = False
            if in_fit:
                cur += val
                launches += 1
        return fits

    f, w = fit_totals("prof_r05_c5_fetch", "FETCH_SIZE"), fit_totals("prof_r05_c5_write", "WRITE_SIZE")
    c5 = {"workload": "C5: 8 tasks, N=Nq=1024, d=512, I=20 (blocked path, csrc/large.h + large_fused.h)", "commit": commit,
          "FETCH_SIZE_KB_per_fit": sum(v for v, _ in f) / len(f) if f else None, "WRITE_SIZE_KB_per_fit": sum(v for v, _ in w) / len(w) if w else None,
          "fits": [len(f), len(w)], "launches_per_fit": f[0][1] if f else None,
          "note": "dispatches of one adkf_fit call by dispatch order (k_lg_begin .. last k_lg_advance): 20 evaluations x (k_lg_build, k_lg_diag of block "
                  "step 0, 8 x (panel, update + sweep of the next diagonal block), k_lg_matvec, k_lg_traces, k_lg_advance); the blocked sweep of Sigma_q "
                  "in the outer stage is not part of it; separate rocprofv3 --pmc passes; KB as reported by rocprofv3, bench.py applies 2 x FETCH_SIZE + "
                  "WRITE_SIZE"}
    with open(os.path.join(OUT, "r05_c5_fit_pmc.json"), "w") as fh:
        json.dump(c5, fh, indent=1)
    print(c5)
    for d in glob.glob(os.path.join(OUT, "prof_r05*")):
        if os.path.isdir(d):
            shutil.rmtree(d, ignore_errors=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "r05":
        return main_r05(sys.argv[2] if len(sys.argv) > 2 else "unknown")
    if len(sys.argv) > 1 and sys.argv[1] == "c5":
        commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
        return main_c5(commit)
    if len(sys.argv) > 1 and sys.argv[1] == "r04":
        commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
        return main_r04(commit)
    commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
    prof = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(OUT, "prof_r03", "**", "*kernel_stats.csv"), recursive=True)
    avg_us = None
    if stats:
        shutil.copy(stats[0], os.path.join(prof, "r03_bench_kernel_stats.csv"))
        avg_us, _ = stats_avg_us(stats[0], KINNER)
    fetch, nf = per_launch("prof_r03_fetch", ["FETCH_SIZE"])
    write, nw = per_launch("prof_r03_write", ["WRITE_SIZE"])
    out = {"kernel": "k_inner<128,512,0,false>", "workload": "C2: 256 tasks, N=Nq=128, d=256, I=20", "commit": commit, "avg_duration_us": avg_us,
           "FETCH_SIZE_KB_per_launch": fetch["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": write["WRITE_SIZE"], "pmc_launches": [nf, nw],
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters alone with --kernel-trace) of `python bench.py --steps 5 --warmup 2`; "
                   "KB as reported, summed over XCDs by rocprofv3; bench.py applies the guide's gfx950 correction (2 x FETCH_SIZE: the D2 reads are "
                   "16-byte-per-lane streams); compulsory traffic of the launch: read D2 (16.8 MB) + write A^-1 (16.8 MB) + vectors"}
    with open(os.path.join(prof, "r03_k_inner_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out)
    sq, n1 = per_launch("prof_r03_sq", ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"])
    lds, n2 = per_launch("prof_r03_lds", ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"])
    sqo = {"kernel": "k_inner<128,512,0,false>", "workload": out["workload"], "commit": commit, "per_launch": {**sq, **{k: v for k, v in lds.items() if k not in sq}},
           "launches": [n1, n2],
           "note": "two rocprofv3 --pmc passes (7 SQ/GRBM counters each); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over "
                   "waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs (MI355X_MICROARCH.md)"}
    if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAIT_ANY") is not None:
        sqo["wait_any_frac_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        sqo["valu_issue_frac_of_wave_cycles"] = (sq.get("SQ_ACTIVE_INST_VALU") or 0.0) / sq["SQ_WAVE_CYCLES"]
    with open(os.path.join(prof, "r03_k_inner_sq_pmc.json"), "w") as fh:
        json.dump(sqo, fh, indent=1)
    print(sqo)
    # configuration 5: the blocked fit is many launches; traffic of ONE fit = sum over the kernels between two adkf_fit calls
    stats5 = glob.glob(os.path.join(OUT, "prof_r03_c5", "**", "*kernel_stats.csv"), recursive=True)
    if stats5:
        shutil.copy(stats5[0], os.path.join(prof, "r03_c5_kernel_stats.csv"))
    fit_kernels = ("k_lg_build", "k_lg_diag", "ProbLgPanel", "ProbLgUpdate", "k_lg_traces", "k_lg_advance", "k_lg_matvec", "k_lg_")
    def fit_total(pattern, counter, fits):
        tot = 0.0
        for row in counter_rows(pattern):
            if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in fit_kernels):
                tot += float(row["Counter_Value"])
        return tot / fits if fits else None
    fits = 3 + 1      # --steps 3 --warmup 1 in tools/round3_profiles.sh
    c5 = {"workload": "C5: 8 tasks, N=Nq=1024, d=512, I=20 (blocked path, csrc/large.h)", "commit": commit,
          "FETCH_SIZE_KB_per_fit": fit_total("prof_r03_c5_fetch", "FETCH_SIZE", fits), "WRITE_SIZE_KB_per_fit": fit_total("prof_r03_c5_write", "WRITE_SIZE", fits),
          "note": "sum over every k_lg_* / ProbLg* launch of the profile (the blocked inner fit AND the blocked sweep of Sigma_q in the outer stage use "
                  "these kernels) divided by the 4 meta-steps of the profiled command; KB as reported by rocprofv3"}
    with open(os.path.join(prof, "r03_c5_fit_pmc.json"), "w") as fh:
        json.dump(c5, fh, indent=1)
    print(c5)


if __name__ == "__main__":
    main()
