#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
K=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_kpre.so
for v in product kpre product kpre; do
  if [ $v = kpre ]; then export MI355TTS_LIB=$K; else unset MI355TTS_LIB; fi
  timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_$v.npz 2>&1 | tail -1
done
python tools/r3/gpt_ab.py --cmp gpurun_out/r3/ab_product.npz gpurun_out/r3/ab_kpre.npz
for v in product kpre product kpre; do
  if [ $v = kpre ]; then export MI355TTS_LIB=$K; else unset MI355TTS_LIB; fi
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done
