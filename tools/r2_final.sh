#!/bin/bash
# round-2 final measurement set (run on the GPU box through gpurun; outputs in gpurun_out/r2final/, copied to profiles/r2/):
#   GPU tests; the default bench line; rocprofv3 kernel stats of the default bench command and of the two secondary
#   configs; short PMC passes (2 DiT evaluations) for F5 fp32 U=1 and bf16 U=8
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r2final; mkdir -p $O
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -6 > $O/tests_gpu.log
timeout 900 python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-secondary --no-cpu-baseline"
R="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 600 $R -d $O/t_f5_f32_u1 -- $B --steps 2 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_f32_under_rocprof.json
timeout 600 $R -d $O/t_f5_bf16_u8 -- $B --dtype bf16 --batch 8 --steps 2 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_bf16_u8_under_rocprof.json
timeout 600 $R -d $O/t_bigvgan_f16_b8 -- $B --workload bigvgan --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_bigvgan_f16_b8_under_rocprof.json
for d in f5_f32_u1 f5_bf16_u8 bigvgan_f16_b8; do cp $O/t_$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/t_*
bash $ROOT/tools/r2_pmc.sh f32 1 > $O/pmc_f32_u1.log 2>&1; cp $ROOT/gpurun_out/r2pmc_f32_u1/pmc_by_kernel.json $O/f5_f32_u1_pmc_by_kernel.json 2>/dev/null
bash $ROOT/tools/r2_pmc.sh bf16 8 > $O/pmc_bf16_u8.log 2>&1; cp $ROOT/gpurun_out/r2pmc_bf16_u8/pmc_by_kernel.json $O/f5_bf16_u8_pmc_by_kernel.json 2>/dev/null
cat $O/tests_gpu.log
python - <<PY
import json
d=json.loads(open("$O/bench_default_final.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
for k in d["roofline"]["kernels"][:6]: print("    ",k["kernel"],round(k["ms_per_step"],3),round(k["avg_launch_us"],1),round(k.get("tflops",0),1))
for n,v in d.get("secondary",{}).items(): print("   sec",n,v["ms_per_step"],v["value"])
print("   cpu",d.get("cpu_baseline"))
PY
head -8 $O/f5_f32_u1_kernel_stats.csv | cut -c1-160
tail -12 $O/pmc_f32_u1.log
