#!/bin/bash
# sample clocks / power while a long GEMM runs
cd $GRAFT_REPO_ROOT
(for i in $(seq 1 16); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' '; echo; sleep 2; done) > gpurun_out/power_trace.txt &
S=$!
sleep 1
ITERS=12000 python tools/gemm_bench.py custom bf16 1 8192 8192 8192 1 1 > gpurun_out/power_gemm.txt 2>&1
MI355TTS_BENCH_ZERO=1 ITERS=6000 python tools/gemm_bench.py custom bf16 1 8192 8192 8192 1 1 >> gpurun_out/power_gemm.txt 2>&1
wait $S
cat gpurun_out/power_gemm.txt; cat gpurun_out/power_trace.txt
