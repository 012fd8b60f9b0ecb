#!/usr/bin/env python
"""Aggregate a rocprofv3 kernel trace by (kernel family, grid) — usage: prof_shapes.py <kernel_trace.csv> [n_forwards]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name']
    if 'dma3' in name: short = 'gemm_dma3'
    elif 'conv_gemm_dma' in name: short = 'gemm_dma2'
    elif 'conv_gemm' in name: short = 'gemm_old'
    elif 'aa_conv' in name: short = 'aa_conv'
    elif 'aa_act' in name: short = 'aa_act'
    elif 'attn' in name: short = 'attn'
    else: short = name[:28]
    key = (short, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k[0]:12s} grid({k[1]},{k[2]},{k[3]}) n={v[0]:5d} avg {v[1]/v[0]:8.1f} us  {100*v[1]/tot:5.1f} %")
print(f"kernel time per forward: {tot/nf/1e3:.2f} ms")
