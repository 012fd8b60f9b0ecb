#!/bin/bash
# round-3 final measurement set (run on the GPU box through gpurun; outputs in gpurun_out/r3final/, copied to profiles/r3/):
#   GPU tests with -rA; the default bench line (all legs incl. PMC traffic, host-io, CPU baseline, secondaries); rocprofv3 kernel
#   stats of the default bench command and of the two secondary configs; short PMC passes (2 DiT evaluations) for F5 fp32 U=1;
#   the BigVGAN / IndexTTS workloads; bench.py --gpus 2 started WITHOUT a launcher (both ranks on this box's one GPU, gloo)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r3final9; mkdir -p $O
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q -x -rA --timeout 900 > $O/tests_gpu_rA.log 2>&1; tail -3 $O/tests_gpu_rA.log
timeout 1200 python bench.py --steps 10 --warmup 2 > $O/bench_default_final.json 2> $O/bench_default_final.err
timeout 600 python bench.py --workload bigvgan > $O/bench_bigvgan_f16_b8.json 2>/dev/null
timeout 600 python bench.py --workload indextts --no-cpu-baseline > $O/bench_indextts.json 2>/dev/null
MI355TTS_BENCH_BACKEND=gloo MI355TTS_BENCH_ONE_GPU=1 MI355TTS_BENCH_SMALL=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_selflaunch_plumbing.json 2> $O/bench_gpus2.err
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-secondary --no-cpu-baseline --no-pmc"
R="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 600 $R -d $O/t_f5_f32_u1 -- $B --steps 3 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_f32_under_rocprof.json
timeout 600 $R -d $O/t_f5_bf16_u8 -- $B --dtype bf16 --batch 8 --steps 2 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_bf16_u8_under_rocprof.json
timeout 600 $R -d $O/t_bigvgan_f16_b8 -- $B --workload bigvgan --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_bigvgan_f16_b8_under_rocprof.json
for d in f5_f32_u1 f5_bf16_u8 bigvgan_f16_b8; do cp $O/t_$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/t_*
# PMC: MFMA busy / wave cycles + fabric traffic per kernel, short command (2 DiT evaluations)
P=$O/pmc; mkdir -p $P
C="python $ROOT/tools/pmc_f5_eval.py f32 1 2"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $P/p_sq -- $C > $P/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/p_fetch -- $C > $P/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/p_write -- $C > $P/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/p_tcc -- $C > $P/tcc.log 2>&1
python $ROOT/tools/pmc_summary.py $P/p_sq $P/p_fetch $P/p_write $P/p_tcc > $O/f5_f32_u1_pmc_by_kernel.json
rm -rf $P
python - <<PY
import json
d=json.loads(open("$O/bench_default_final.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["roofline"]["traffic"])
for k in d["roofline"]["kernels"][:6]: print("    ",k["kernel"],round(k["ms_per_step"],3),round(k["avg_launch_us"],1),round(k.get("tflops",0),1))
for n,v in d.get("secondary",{}).items(): print("   sec",n,round(v["ms_per_step"],1),round(v["value"],1))
print("   cpu",d.get("cpu_baseline",{}).get("value"), "host-io", d["config"].get("host_io_ms_per_step"))
p=json.load(open("$O/f5_f32_u1_pmc_by_kernel.json"))
for k,v in list(p.items())[:4]:
    print(k[:70], {c:(round(v[c]["per_dispatch"],1) if isinstance(v.get(c),dict) else v.get(c)) for c in ("SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CU_CYCLES","FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum","TCC_MISS_sum")}, v.get("hbm_GB_corrected_per_dispatch"))
PY
head -6 $O/f5_f32_u1_kernel_stats.csv | cut -c1-170
for f in bench_bigvgan_f16_b8 bench_indextts bench_gpus2_selflaunch_plumbing; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), round(d['value'],1), d['n_gpus'])"; done
