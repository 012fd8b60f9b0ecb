#!/usr/bin/env python3
"""Average duration of the four DiT linear layers (QKV, O, FF1, FF2) from a rocprofv3 --kernel-trace CSV of an F5 fp32 run: the
panel-plane GEMM launches of a block follow each other in that order, so launch i of the kernel is layer i % 4.

    python tools/x3p_by_shape.py <dir with *kernel_trace.csv> [name-substring, default linear_x3p_kernel]
"""
import csv, glob, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "linear_x3p_kernel"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
names = ["QKV", "O", "FF1", "FF2"]
acc = [[0, 0] for _ in range(4)]
for i, (_, dur) in enumerate(rows):
    acc[i % 4][0] += dur; acc[i % 4][1] += 1
print(f"{len(rows)} launches of *{sub}*")
for n, (t, c) in zip(names, acc):
    print(f"  {n:4s} {t / max(c, 1) / 1e3:8.2f} us  x {c}")
