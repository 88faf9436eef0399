#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (kernel-trace --stats pass + FETCH_SIZE pass + WRITE_SIZE pass of ONE bench.py command, as
collected by profiles/collect.sh) into the small files that are committed:

  <tag>_kernel_stats.csv        rocprofv3's own summary, verbatim (every launch of the process)
  <tag>_timed_kernel_stats.csv  the same columns recomputed from the kernel-trace CSV over the TIMED launches only: the last
                                `roofline.launches` dispatches of the dominant kernel (bench.py runs its warm-up runs first and
                                nothing after the timed runs when called with --no-extras) and every other kernel dispatched
                                from the first of them on.  This is the average that must agree with bench.py's HIP-event
                                `roofline.avg_launch_ms`.
  <tag>_pmc.json                per-kernel means of the counters (gfx950 correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE
                                under-reports wide coalesced reads by 2x; both counters are in KiB), the bench lines of the
                                three passes, and the cross-check of the two averages.

    python profiles/summarize.py gpurun_out/prof_sunspot64 r02a_sunspot64 [--current sunspot64]

--current NAME also writes profiles/current_pmc_NAME.json, the copy bench.py reads for `roofline.traffic`.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except Exception:
                continue
    return None


def timed_stats(trace_csv, kernel_stem, n_timed):
    rows = []
    for r in csv.DictReader(open(trace_csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    dom = [k for k, r in enumerate(rows) if kernel_stem in r[2]]
    if not dom or n_timed <= 0 or n_timed > len(dom):
        return None, None
    first = rows[dom[-n_timed]][0]
    agg = collections.defaultdict(list)
    for st, en, name in rows:
        if st >= first:
            agg[name].append(en - st)
    total = sum(sum(v) for v in agg.values())
    out = []
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append({"Name": name, "Calls": len(v), "TotalDurationNs": sum(v), "AverageNs": sum(v) / len(v),
                    "Percentage": 100.0 * sum(v) / total, "MinNs": min(v), "MaxNs": max(v)})
    dom_avg = next(e["AverageNs"] for e in out if kernel_stem in e["Name"])
    return out, dom_avg


def main():
    src, tag = sys.argv[1], sys.argv[2]
    current = sys.argv[sys.argv.index("--current") + 1] if "--current" in sys.argv else None
    here = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(here, f"{tag}_kernel_stats.csv"))
    out = {"tag": tag, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, + --kernel-trace --stats pass",
           "commit": os.environ.get("PTNN_COMMIT"),         # the commit the three passes ran (exported by whoever calls collect.sh; the GPU box has no .git)
           "kernels": {}}
    for j in ("kt", "fetch", "write"):
        p = os.path.join(src, j + ".log")
        if os.path.exists(p):
            line = last_json_line(p)
            if line is not None:
                out["bench_" + j] = line
    cmd = os.path.join(src, "command.txt")
    if os.path.exists(cmd):
        out["command"] = open(cmd).read().strip()
    # ---- timed launches only, from the kernel trace of the --stats pass
    bench = out.get("bench_kt")
    if bench:
        out["mh_steps_per_launch"] = bench.get("roofline", {}).get("mh_steps_per_launch")     # what "per launch" means in this file
    tr = glob.glob(os.path.join(src, "kt", "**", "*_kernel_trace.csv"), recursive=True)
    if bench and tr:
        roof = bench["roofline"]
        stem = roof["kernel"].split("::")[-1].split("<")[0] + "<"
        rows, dom_avg = timed_stats(tr[0], stem, int(roof["launches"]))
        if rows:
            with open(os.path.join(here, f"{tag}_timed_kernel_stats.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0]))
                w.writeheader()
                w.writerows(rows)
            out["cross_check"] = {"kernel": roof["kernel"], "timed_launches": int(roof["launches"]),
                                  "rocprof_avg_ms_timed_launches": dom_avg * 1e-6,
                                  "bench_hip_event_avg_launch_ms": roof["avg_launch_ms"],
                                  "bench_ms_per_step": bench["ms_per_step"],
                                  "roofline_frac_recomputed": roof["algorithmic_bytes_per_launch"] / (dom_avg * 1e-9) / 1e9 / roof["peak"]}
    for name in ("fetch", "write"):
        fs = glob.glob(os.path.join(src, name, "**", "*_counter_collection.csv"), recursive=True)
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            agg[(r["Kernel_Name"], r["Counter_Name"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                 r["Workgroup_Size"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        for (k, c, vg, sg, lds, wg, grid), v in agg.items():
            if "ptnn" not in k:
                continue
            e = out["kernels"].setdefault(k, {"launches": len(v), "vgpr": int(vg), "sgpr": int(sg), "lds_bytes": int(lds),
                                              "workgroup": int(wg), "grid": int(grid)})
            e[c + "_KiB_mean"] = sum(v) / len(v)
    for k, e in out["kernels"].items():
        if "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024.0
    json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
    if current:
        json.dump(out, open(os.path.join(here, f"current_pmc_{current}.json"), "w"), indent=1)
    print(open(os.path.join(here, f"{tag}_kernel_stats.csv")).read())
    print(json.dumps(out.get("cross_check"), indent=1))
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
