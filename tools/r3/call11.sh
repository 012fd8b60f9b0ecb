#!/bin/bash
# per-kernel durations of the decode step, base vs new
mkdir -p gpurun_out/r3
cd /root/repo
export TMPDIR=/tmp
B=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_base.so
for v in base new; do
  if [ $v = base ]; then export MI355TTS_LIB=$B; else unset MI355TTS_LIB; fi
  rm -rf /tmp/prof_$v
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p --output-format csv -- python tools/r3/gpt_ab.py /tmp/x_$v.npz > /tmp/log_$v.txt 2>&1
  tail -1 /tmp/log_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  cp $f gpurun_out/r3/gpt_decode_${v}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:9]: print(r['Name'][:120], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
