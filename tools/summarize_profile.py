#!/usr/bin/env python3
"""Condense a gpurun_out/<run>/ rocprofv3 series into the committed summaries under profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg / min / max)
  profiles/<tag>_pmc.json           per-launch PMC values of the dominant kernel (FETCH_SIZE, WRITE_SIZE, SQ_*)
  profiles/pmc_<workload>.json      {"hbm_bytes_per_launch": ...} read by bench.py for roofline.traffic

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE come from separate
passes, are in KiB, and on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x (16 B/lane;
our loads are 8 B/lane, "uncalibrated": both the raw and the doubled figure are recorded, the doubled one is
the conservative number used for traffic).
Usage: python tools/summarize_profile.py gpurun_out/r01e r01e c2 "k_indirect<14"
       python tools/summarize_profile.py gpurun_out/r01e r01e c2_ndim12 "k_indirect<12" c2    (12-dim leg of the same trace)
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def main():
    run, tag, workload, kname = sys.argv[1:5]
    src = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else workload      # prof_<src>/ holds the kernel trace (one trace, several kernels)
    # <run>/<pmc_src>/pmc_* hold the counter passes: the workload's own directory when there is one (bench.py's pmc_key), else the trace's
    pmc_src = workload if os.path.isdir(os.path.join(run, workload)) else src
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    # kernel stats
    stats = list(rows(os.path.join(run, "prof_%s" % src, "**", "*_kernel_stats.csv")))
    out_csv = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, workload))
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in stats:
            name = r["Name"]
            if len(name) > 120:
                name = name[:117] + "..."
            w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    dom = [r for r in stats if kname in r["Name"]]
    summary = {"run": run, "workload": workload, "kernel": dom[0]["Name"] if dom else None,
               "kernel_trace": {"calls": int(dom[0]["Calls"]), "avg_ns": float(dom[0]["AverageNs"]), "min_ns": float(dom[0]["MinNs"]),
                                "max_ns": float(dom[0]["MaxNs"])} if dom else None, "pmc": {}}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        acc = {}
        for r in list(rows(os.path.join(run, sub, "**", "*_counter_collection.csv"))) + list(rows(os.path.join(run, pmc_src, sub, "**", "*_counter_collection.csv"))):
            if kname not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            summary["pmc"].setdefault("_meta", {"VGPR_Count": r["VGPR_Count"], "Accum_VGPR_Count": r["Accum_VGPR_Count"],
                                                 "SGPR_Count": r["SGPR_Count"], "Scratch_Size": r["Scratch_Size"],
                                                 "Grid_Size": r["Grid_Size"], "Workgroup_Size": r["Workgroup_Size"]})
        for k, v in acc.items():
            summary["pmc"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
    fetch_kib = summary["pmc"].get("FETCH_SIZE", {}).get("per_launch_mean")
    write_kib = summary["pmc"].get("WRITE_SIZE", {}).get("per_launch_mean")
    # helper kernels that belong to the same sweep (argument 6: "k_node_records+k_pack"): their bytes per launch of the dominant
    # kernel (launch counts of the same counter pass) are added -- the staged sweeps of round 4 are three kernels, not one
    helpers = sys.argv[6].split("+") if len(sys.argv) > 6 else []
    if helpers and fetch_kib is not None and write_kib is not None:
        summary["helpers"] = {}
        for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
            per = {}
            n_dom = 0
            for r in list(rows(os.path.join(run, sub, "**", "*_counter_collection.csv"))) + list(rows(os.path.join(run, pmc_src, sub, "**", "*_counter_collection.csv"))):
                if r["Counter_Name"] != counter:
                    continue
                if kname in r["Kernel_Name"]:
                    n_dom += 1
                for h in helpers:
                    if h in r["Kernel_Name"]:
                        per.setdefault(h, []).append(float(r["Counter_Value"]))
            for h, v in per.items():
                add = sum(v) / max(n_dom, 1)
                summary["helpers"].setdefault(h, {})[counter + "_kib_per_sweep"] = add
                if counter == "FETCH_SIZE":
                    fetch_kib += add
                else:
                    write_kib += add
        summary["pmc"]["FETCH_SIZE"]["per_sweep_with_helpers"] = fetch_kib
        summary["pmc"]["WRITE_SIZE"]["per_sweep_with_helpers"] = write_kib
    if fetch_kib is not None and write_kib is not None:
        summary["hbm_bytes_per_launch_raw"] = (fetch_kib + write_kib) * 1024.0
        summary["hbm_bytes_per_launch"] = (2.0 * fetch_kib + write_kib) * 1024.0
        with open(os.path.join(ROOT, "profiles", "pmc_%s.json" % workload), "w") as fh:
            json.dump({"hbm_bytes_per_launch": summary["hbm_bytes_per_launch"], "hbm_bytes_per_launch_raw": summary["hbm_bytes_per_launch_raw"],
                       "fetch_kib": fetch_kib, "write_kib": write_kib, "source": "%s (%s)" % (run, tag)}, fh, indent=1)
    if not any(k != "_meta" for k in summary["pmc"]):
        raise SystemExit("no counter rows for kernel %r under %s/%s/pmc_*: nothing written (an empty pmc object is not evidence)" % (kname, run, pmc_src))
    # issue occupancy x useful-slot fraction, the decomposition DESIGN section 6 quotes per row
    sq = summary["pmc"]
    if "SQ_INSTS_VALU" in sq and "SQ_BUSY_CYCLES" in sq:
        summary["valu_wave_instructions_per_launch"] = sq["SQ_INSTS_VALU"]["per_launch_mean"]
    with open(os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (tag, workload)), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
