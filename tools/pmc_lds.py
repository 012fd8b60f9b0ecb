#!/usr/bin/env python
"""Aggregate a rocprofv3 `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` counter_collection.csv by kernel:
share of LDS-active cycles lost to bank conflicts.  usage: pmc_lds.py <counter_collection.csv>"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    for tag in ("conv_gemm_dma3", "conv_gemm_dma_kernel", "conv_gemm_kernel", "aa_conv", "aa_act", "attn_kernel", "gpt_attn",
                "gemv", "gemm_skinny", "rownorm", "conv_post"):
        if tag in name:
            name = tag
            break
    agg[name[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = [(k, v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_IDX_ACTIVE"]) for k, v in agg.items() if v["SQ_LDS_IDX_ACTIVE"] > 0]
tot = sum(a for _, _, a in rows)
for k, c, a in sorted(rows, key=lambda t: -t[2]):
    print(f"{k:40s} conflict / active = {c / a:6.3f}   share of all LDS-active cycles {100 * a / tot:5.1f} %")
