"""Static instruction census of one kernel in a hipcc -save-temps .s file: totals per class and per mnemonic.
Usage: python tools/isa_count.py FILE.s KERNEL_SUBSTRING [top]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ops = collections.Counter()
for l in lines[start + 1:end]:
    l = l.split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."):
        continue
    ops[l.split()[0]] += 1
total = sum(ops.values())
cls = collections.Counter()
for op, n in ops.items():
    if op.startswith("v_") and ("f64" in op):
        cls["valu fp64"] += n
    elif op.startswith("v_pk_"):
        cls["valu packed"] += n
    elif op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        cls["valu transcendental"] += n
    elif op.startswith("v_"):
        cls["valu other"] += n
    elif op.startswith("scratch_"):
        cls["scratch"] += n
    elif op.startswith(("global_", "buffer_", "flat_")):
        cls["vmem"] += n
    elif op.startswith("ds_"):
        cls["lds"] += n
    elif op.startswith("s_waitcnt"):
        cls["s_waitcnt"] += n
    elif op.startswith(("s_cbranch", "s_branch")):
        cls["branch"] += n
    elif op.startswith("s_load") or op.startswith("s_buffer"):
        cls["smem"] += n
    elif op.startswith("s_"):
        cls["salu"] += n
    else:
        cls["other"] += n
print(f"{key}: {total} instructions, lines {start}-{end}")
for k, n in cls.most_common():
    print(f"  {k:22s} {n:6d}")
for op, n in ops.most_common(top):
    print(f"    {op:32s} {n:5d}")
