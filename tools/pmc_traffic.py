#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter CSVs (one pass per counter, as MI355X_MICROARCH.md §HBM prescribes) into HBM-side
bytes per forward by kernel family.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- python bench.py --steps S --warmup W --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- python bench.py --steps S --warmup W --no-cpu-baseline
    python tools/pmc_traffic.py out/fetch out/write FORWARDS > traffic.json      (FORWARDS = S + W)

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
(the guide's calibration): `fetch_GB_corrected` = 2 x raw.  WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys


def family(name: str) -> str:
    if "conv_gemm_dma3" in name: return "gemm_dma3"
    if "conv_gemm_dma_kernel" in name: return "gemm_dma2"
    if "conv_gemm_kernel" in name: return "gemm_regstaged"
    if "aa_conv" in name: return "aa_conv"
    if "aa_act" in name: return "aa_act"
    if "gemv" in name: return "gemv"
    if "attn" in name: return "attn"
    if "conv_post" in name: return "conv_post"
    return "other"


def read(d: str, counter: str):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    tot, cnt = collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            fam = family(r["Kernel_Name"])
            tot[fam] += float(r["Counter_Value"])
            cnt[fam] += 1
    return tot, cnt


def main():
    fdir, wdir, forwards = sys.argv[1], sys.argv[2], int(sys.argv[3])
    ft, fc = read(fdir, "FETCH_SIZE")
    wt, wc = read(wdir, "WRITE_SIZE")
    out = {}
    for fam in sorted(set(ft) | set(wt)):
        out[fam] = {"launches_per_forward": fc[fam] / forwards,
                    "fetch_GB_raw": ft[fam] * 1024 / forwards / 1e9,
                    "fetch_GB_corrected": 2 * ft[fam] * 1024 / forwards / 1e9,
                    "write_GB": wt[fam] * 1024 / forwards / 1e9}
    conv = [f for f in out if f.startswith("gemm") or f == "aa_conv"]
    out["_conv_family"] = {"launches_per_forward": sum(out[f]["launches_per_forward"] for f in conv),
                           "traffic_GB_per_forward": sum(out[f]["fetch_GB_corrected"] + out[f]["write_GB"] for f in conv)}
    out["_all"] = {"traffic_GB_per_forward": sum(v["fetch_GB_corrected"] + v["write_GB"] for k, v in out.items() if not k.startswith("_"))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
