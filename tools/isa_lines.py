"""Attributes the instructions of one kernel in a `hipcc -gline-tables-only -save-temps` .s file to source lines
(.loc directives) and prints the count per (file, line range bucket).
Usage: python tools/isa_lines.py FILE.s KERNEL_SUBSTRING [bucket]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
cur = ("?", 0)
cnt = collections.Counter()
valu = collections.Counter()
for l in lines[start + 1:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)) // bucket * bucket)
        continue
    t = l.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        continue
    cnt[cur] += 1
    if t.startswith("v_"):
        valu[cur] += 1
for (f, ln), n in sorted(cnt.items()):
    print(f"{f}:{ln:5d}  all {n:5d}  valu {valu[(f, ln)]:5d}")
print("total", sum(cnt.values()), "valu", sum(valu.values()))
