#!/bin/bash
# A/B of two builds of the library on one box: tools/ab/lib_old.so vs tools/ab/lib_new.so (MI355TTS_LIB), alternating
cd "$(dirname "$0")/../.."
SUM='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), [(k["kernel"][:44],round(k["avg_launch_us"],1)) for k in d["roofline"]["kernels"][:5]])'
for rep in 1 2; do for v in old new; do
  echo "== $v bf16 U=8"; MI355TTS_LIB=$PWD/tools/ab/lib_$v.so python bench.py --dtype bf16 --batch 8 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "$SUM"
done; done
for v in old new; do
  echo "== $v f32 U=1"; MI355TTS_LIB=$PWD/tools/ab/lib_$v.so python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "$SUM"
done
