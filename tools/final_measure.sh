#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun); writes gpurun_out/final/.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tail -1 > $O/bench_bigvgan_f16_b8.json
python bench.py --dtype f32 --batch 1 --steps 10 2>/dev/null | tail -1 > $O/bench_bigvgan_f32_b1.json
python bench.py --workload f5 2>/dev/null | tail -1 > $O/bench_f5_bf16_u1.json
python bench.py --workload f5 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f5_bf16_u8.json
python bench.py --workload f5 --dtype f32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f5_f32_u1.json
python bench.py --workload indextts_f --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_indextts_f_f16.json
python bench.py --workload indextts 2>/dev/null | tail -1 > $O/bench_indextts_f16_b1.json
python bench.py --workload indextts --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_indextts_f16_b8.json
python bench.py --workload indextts --batch 16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_indextts_f16_b16.json
cd /tmp; export TMPDIR=/tmp
R="rocprofv3 --kernel-trace --stats --output-format csv"
$R -d $O/prof_bigvgan -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bigvgan_under_rocprof.json
$R -d $O/prof_f5u1 -- python $GRAFT_REPO_ROOT/bench.py --workload f5 --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f5_u1_under_rocprof.json
$R -d $O/prof_f5u8 -- python $GRAFT_REPO_ROOT/bench.py --workload f5 --batch 8 --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f5_u8_under_rocprof.json
$R -d $O/prof_ix -- python $GRAFT_REPO_ROOT/bench.py --workload indextts --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_indextts_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 3 > $O/bigvgan_pmc_hbm_traffic.json
# keep only the small summaries (kernel traces are tens of MB)
for d in prof_bigvgan prof_f5u1 prof_f5u8 prof_ix; do cp $O/$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/prof_bigvgan $O/prof_f5u1 $O/prof_f5u8 $O/prof_ix $O/pmc_fetch $O/pmc_write
ls -la $O
