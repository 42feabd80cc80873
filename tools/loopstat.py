#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing: python tools/loopstat.py file.s <mangled-substring>
For every backward branch (label .. branch) prints the span's VALU / SALU / VMEM / LDS / f64 counts, innermost-first."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E", "E:")) or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


loops = []
for i, l in enumerate(body):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))
loops.sort(key=lambda t: t[1] - t[0])
for a, b in loops:
    cnt = {}
    f64 = 0
    ops = {}
    for l in body[a:b + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m or l.strip().startswith((".", ";")):
            continue
        op = m.group(1)
        k = kind(op)
        cnt[k] = cnt.get(k, 0) + 1
        if "f64" in op:
            f64 += 1
        if k in ("vmem", "lds"):
            ops[op] = ops.get(op, 0) + 1
    if cnt.get("vmem", 0) == 0 and len(sys.argv) < 4:
        continue
    print("lines %6d..%6d  valu %4d (f64 %3d) salu %4d wait %3d vmem %3d lds %3d  %s" % (
        start + a + 1, start + b + 1, cnt.get("valu", 0), f64, cnt.get("salu", 0), cnt.get("wait", 0), cnt.get("vmem", 0), cnt.get("lds", 0),
        " ".join("%s:%d" % kv for kv in sorted(ops.items()))))
