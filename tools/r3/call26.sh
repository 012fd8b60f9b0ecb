#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_r.npz 2>&1 | tail -1
rm -rf /tmp/prof_new
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_new -o p --output-format csv -- python tools/r3/gpt_ab.py /tmp/x_new.npz > /tmp/log_new.txt 2>&1
f=$(find /tmp/prof_new -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r3/gpt_decode_r_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:8]: print(r['Name'][:110], r['Calls'], r['AverageNs'], r['Percentage'])
PY
timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compat.py -x -q -m gpu 2>&1 | tail -5
