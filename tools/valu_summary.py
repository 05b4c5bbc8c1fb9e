"""VALU figures per kernel from the PMC passes of tools/profile_gpu.sh, as JSON for bench.py (roofline.valu_measured).

  python tools/valu_summary.py gpurun_out/prof_TAG > profiles/rNN_valu.json

Per kernel: wave-instructions per launch (SQ_INSTS_VALU), average duration, and the share of the chip's VALU issue
slots they fill: a wave64 instruction occupies a 32-lane SIMD for 2 cycles, the chip has 1024 SIMDs, and the kernel's
cycles are taken from SQ_BUSY_CYCLES (summed over the 32 shader engines by the counter, hence / 32)."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


dur = {}
for f in glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
        if k in dur:  # several instantiations of one template: weight by calls
            c0, a0 = dur[k]
            dur[k] = (c0 + calls, (a0 * c0 + avg * calls) / (c0 + calls))
        else:
            dur[k] = (calls, avg)
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    if k not in dur or "SQ_INSTS_VALU" not in c:
        continue
    insts = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])
    busy = sum(c["SQ_BUSY_CYCLES"]) / len(c["SQ_BUSY_CYCLES"]) / 32.0 if c.get("SQ_BUSY_CYCLES") else None
    e = {"calls": dur[k][0], "avg_us": round(dur[k][1], 1), "valu_wave_instructions_per_launch": round(insts),
         "waves_per_launch": round(sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])) if c.get("SQ_WAVES") else None}
    if busy:
        e["valu_issue_slot_share"] = round(insts * 2.0 / (1024.0 * busy), 4)
    # against the rate tools/experiments/valu_rate.hip measures on this chip with two or more waves per SIMD: one
    # fp32 wave-instruction per 1.2 ns and SIMD (packed fp32 no faster per element, reciprocals 3x slower)
    e["valu_share_of_measured_rate"] = round(insts * 1.2e-3 / 1024.0 / dur[k][1], 4)
    out[k] = e
json.dump(out, sys.stdout, indent=1, sort_keys=True)
print()
