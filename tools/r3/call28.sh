#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_novt.so
for d in 0 32 0 32; do
  rm -rf /tmp/pf; cd /tmp; MI355TTS_GEMM_DBG=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python /root/repo/bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$d', d['ms_per_step'])"
  cd /root/repo
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pf/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:2]: print('   ', r['Name'][:60], r['Calls'], r['AverageNs'])
PY
done
