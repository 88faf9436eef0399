#!/usr/bin/env python3
"""MFMA utilisation of the dominant kernel from ONE rocprofv3 counter pass (on the GPU box):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
              GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload W ... --no-extras
    python3 profiles/tools/mfma_pmc.py DIR KERNEL_STEM out.json

Per launch of the kernel (means over its dispatches): the raw counters, MFMA flops = MOPS_F32 x 512 (the counter's own unit),
matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) (busy cycles are summed over the SIMDs,
GRBM_GUI_ACTIVE over the 8 XCDs: divided by the launch duration it gives 8 x the shader clock), and the flop rate from the kernel
trace's duration.  No gfx950 section exists in the shipped derived-counter files (MI355X guide), so
the raw counters are combined here with the formula stated."""
import collections
import csv
import glob
import json
import sys

d, stem, out = sys.argv[1], sys.argv[2], sys.argv[3]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = set()
for r in csv.DictReader(open(cc[0])):
    if stem in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        disp.add(r["Dispatch_Id"])
n = len(disp)
mean = collections.defaultdict(float)
for k in acc.values():
    for c, v in k.items():
        mean[c] += v / n
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if stem in r["Kernel_Name"]] if kt else []
res = {"kernel_stem": stem, "launches": n, "counters_per_launch_mean": dict(mean)}
if dur:
    res["avg_launch_ms_in_this_pass"] = sum(dur) / len(dur) / 1e6
flops = mean.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0
res["mfma_flops_per_launch"] = flops
bflops = mean.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512.0     # the split-operand forward pass runs on the bf16 instruction
if bflops:
    res["mfma_bf16_flops_per_launch"] = bflops
    if dur:
        res["mfma_bf16_tflops"] = bflops / (sum(dur) / len(dur) * 1e-9) / 1e12
        res["mfma_bf16_frac_of_dense_peak_2500"] = res["mfma_bf16_tflops"] / 2500.0
if dur and flops:
    res["mfma_tflops"] = flops / (sum(dur) / len(dur) * 1e-9) / 1e12
    res["mfma_frac_of_fp32_matrix_peak_157.3"] = res["mfma_tflops"] / 157.3
if mean.get("GRBM_GUI_ACTIVE"):
    res["matrix_pipe_busy_frac"] = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    res["formula"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)"
    if dur:
        res["shader_clock_ghz"] = mean["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / len(dur))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
