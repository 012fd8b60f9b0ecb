#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_gpu_f5.py tests/test_gpu_gpt.py tests/test_gpu_compat.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r3/bench_c22.json 2> gpurun_out/r3/bench_c22.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_c22.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"])
for k in d["roofline"].get("kernels", [])[:3]: print(k["kernel"][:70], round(k["avg_launch_us"],1), k["launches_per_step"], round(k["ms_per_step"],1))
for n,s in d.get("secondary",{}).items(): print(n, round(s["ms_per_step"],1), round(s["value"],1))
PY
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python /root/repo/bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pf/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
