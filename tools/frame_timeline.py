"""Timeline of the last complete frame in a rocprofv3 kernel trace: python tools/frame_timeline.py <dir-with-kernel_trace.csv>
Prints every kernel / copy / fill with its start, duration and the idle gap before it."""
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
f = max(fs, key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_chunk_cull"]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end, gaps, busy = None, 0.0, 0.0
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-30s start %8.1f dur %7.1f gap %5.1f" % (names[i][:30], (s - t0) / 1e3, (e - s) / 1e3, gap))
    gaps += max(gap, 0.0); busy += (e - s) / 1e3
    prev_end = e
print("busy %.1f us, idle %.1f us, period %.1f us" % (busy, gaps, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
