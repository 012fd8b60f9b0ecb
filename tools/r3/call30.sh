#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f5.py -x -q -m gpu -rA -k "bit_neutral or full_size_fp32" 2>&1 | grep -E "DiT evaluation|F5 full size|passed|failed"
for v in occ2 new occ2 new; do
  if [ $v = occ2 ]; then export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_occ2.so; else unset MI355TTS_LIB; fi
  rm -rf /tmp/pf; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python /root/repo/bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  cd /root/repo
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pf/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[1:2]: print('   ', r['Name'][:60], r['Calls'], r['AverageNs'])
PY
done
