"""Summarises rocprofv3 output directories written by tools/profile_gpu.sh.

  python tools/pmc_summary.py gpurun_out/prof_TAG [kernel-name-substring ...]

Per kernel: average duration (kernel-trace stats) and the mean of every PMC counter per dispatch.
SQ_* cycle counters are in quad-cycles. FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 prints them; FETCH_SIZE
tallies every 128-byte line fetch as 64 bytes on gfx950, whatever the access shape (tools/fetch_calib.hip): bytes read
= 2 x FETCH_SIZE x 1024.
"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
only = sys.argv[2:]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


dur = {}
for f in glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
acc = defaultdict(lambda: defaultdict(list))
regs = {}
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        regs[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
for k in sorted(dur, key=lambda k: -dur[k][2]):
    if only and not any(o in k for o in only):
        continue
    calls, avg, pct = dur[k]
    print(f"{k}: calls {calls}, avg {avg:.1f} us, {pct:.2f} % of GPU time; vgpr/sgpr/lds/scratch = {regs.get(k)}")
    for c, v in sorted(acc.get(k, {}).items()):
        print("   %-24s n=%-4d mean=%.5g" % (c, len(v), sum(v) / len(v)))
