#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for r in 16 8 16 8; do
  rm -rf /tmp/pf; cd /tmp; MI355TTS_AA_R=$r timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python /root/repo/bench.py --workload bigvgan --no-secondary --no-cpu-baseline --no-pmc --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('R=$r', d['ms_per_step'])"
  cd /root/repo
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pf/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:5]:
    if 'aa_act' in r['Name']: print('   ', r['Name'][:70], r['Calls'], r['AverageNs'])
PY
done
