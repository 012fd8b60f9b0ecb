#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_f5.py tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -3
rm -rf /tmp/pf; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python /root/repo/bench.py --dtype bf16 --batch 8 --no-secondary --no-cpu-baseline --no-pmc --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 u8 step', d['ms_per_step'])"
cd /root/repo
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pf/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]: print('   ', r['Name'][:80], r['Calls'], r['AverageNs'])
PY
