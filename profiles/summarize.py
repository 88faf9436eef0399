#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (kernel-trace --stats pass + FETCH_SIZE pass + WRITE_SIZE pass, as collected by
the commands in profiles/README.md) into the two small files that are committed: <tag>_kernel_stats.csv (verbatim
rocprofv3 summary) and <tag>_pmc.json (per-kernel means of the counters, with the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md section HBM applied: FETCH_SIZE under-reports wide coalesced reads by 2x; both counters are in KiB).

    python profiles/summarize.py gpurun_out/prof1 r01_sunspot64_lg
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(here, f"{tag}_kernel_stats.csv"))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, + --kernel-trace", "kernels": {}}
    for name in ("fetch", "write"):
        f = glob.glob(os.path.join(src, name, "**", "*_counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
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
    for j in ("bench_kt.json", "bench_fetch.json", "bench_write.json"):
        p = os.path.join(src, j)
        if os.path.exists(p):
            out[j] = json.loads(open(p).read().strip().splitlines()[-1])
    json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
    print(open(os.path.join(here, f"{tag}_kernel_stats.csv")).read())
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
