#!/bin/bash
# round 3, GPU call 1: aa_conv determinism A/B, BigVGAN + multirank tests on the new AA form / launcher, default bench
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
bash tools/ubench/aa_race.sh run > $O/aa_race.txt 2>&1
tail -5 $O/aa_race.txt
timeout 1200 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_multirank.py -x -q -rA -m gpu > $O/pytest_c1.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_c1.log
timeout 900 python bench.py > $O/bench_c1.json 2> $O/bench_c1.err; echo "bench rc=$?"
tail -c 1500 $O/bench_c1.json; tail -5 $O/bench_c1.err
