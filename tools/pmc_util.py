#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc passes (counter_collection.csv files) by kernel family: counter sums and ratios to
SQ_BUSY_CU_CYCLES.  usage: pmc_util.py <csv> [<csv> ...]"""
import collections, csv, sys
TAGS = ("conv_gemm_dma3", "conv_gemm_dma_kernel", "conv_gemm_kernel", "aa_conv", "aa_act", "attn_kernel", "rownorm", "conv_post")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for t in TAGS:
            if t in name:
                name = t
                break
        else:
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
    busy = v.get("SQ_BUSY_CU_CYCLES", 0) or 1.0
    print(k)
    for c in names:
        if c in v:
            print(f"    {c:30s} {v[c]:14.4g}   / busy_cu_cycles = {v[c] / busy:8.3f}")
