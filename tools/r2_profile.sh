#!/bin/bash
# round-2 profile set: rocprofv3 kernel stats + PMC passes of the DEFAULT bench command (F5-TTS fp32 NFE=32, configs[2]),
# plus kernel stats of the two secondary configs.  Run on the GPU box through gpurun; summaries land in gpurun_out/r2prof/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r2prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-secondary --no-cpu-baseline"
R="rocprofv3 --kernel-trace --stats --output-format csv"
$R -d $O/t_f5f32 -- $B --steps 2 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_f32_under_rocprof.json
$R -d $O/t_f5u8 -- $B --dtype bf16 --batch 8 --steps 2 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_bf16_u8_under_rocprof.json
$R -d $O/t_bv -- $B --workload bigvgan --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_bigvgan_f16_b8_under_rocprof.json
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$c -- $B --steps 1 --warmup 2 > /dev/null 2>&1
done
for d in t_f5f32 t_f5u8 t_bv; do cp $O/$d/*/*kernel_stats.csv $O/${d#t_}_kernel_stats.csv 2>/dev/null; done
python $ROOT/tools/pmc_summary.py $O/p_SQ_VALU_MFMA_BUSY_CYCLES $O/p_SQ_BUSY_CU_CYCLES $O/p_FETCH_SIZE $O/p_WRITE_SIZE > $O/f5_f32_pmc_by_kernel.json
rm -rf $O/t_* $O/p_*
ls -la $O; head -12 $O/f5f32_kernel_stats.csv
