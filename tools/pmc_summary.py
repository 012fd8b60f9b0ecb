#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 --pmc passes (one counter per pass, each pass its own directory).

    python tools/pmc_summary.py <dir> [<dir> ...]  > by_kernel.json

Per kernel name (template arguments kept, argument list dropped): dispatches seen, and for every counter its sum and
its per-dispatch mean.  FETCH_SIZE / WRITE_SIZE are in KB; `hbm_GB_corrected` applies the gfx950 calibration of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-byte request: x2; WRITE_SIZE as reported)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("mi::", "")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
    out = {}
    for k, v in agg.items():
        e = {}
        for c, s in v.items():
            e[c] = {"sum": s, "dispatches": cnt[k][c], "per_dispatch": s / max(cnt[k][c], 1)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CU_CYCLES" in v and v["SQ_BUSY_CU_CYCLES"] > 0:
            e["mfma_busy_over_busy_cu_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CU_CYCLES"]
        if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
            n = max(cnt[k].get("FETCH_SIZE", 0), cnt[k].get("WRITE_SIZE", 0), 1)
            e["hbm_GB_corrected_per_dispatch"] = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / n / 1e9
        out[k] = e
    order = sorted(out, key=lambda k: -out[k].get("SQ_BUSY_CU_CYCLES", {}).get("sum", 0.0))
    json.dump({k: out[k] for k in order}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
