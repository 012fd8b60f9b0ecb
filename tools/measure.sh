#!/bin/bash
# One measurement driver for the GPU box (run it through gpurun from the repo root; outputs under gpurun_out/$ROUND/<what>/):
#
#   tools/measure.sh tests [pytest args]        the -m gpu suite with -rA
#   tools/measure.sh bench [bench.py args]      one bench line (+ a short readable summary)
#   tools/measure.sh ab '<args A>' '<args B>' [n]   same-box A/B of two bench.py command lines, n alternations (default 2)
#   tools/measure.sh layers [bench.py args]     average time of the four DiT linear layers (rocprofv3 --kernel-trace + tools/x3p_by_shape.py)
#   tools/measure.sh cpu-full                   the CPU baseline with all 31 evaluations at 8 threads (no GPU work)
#   tools/measure.sh final                      the round-end set: GPU tests, default line (PMC traffic, host I/O, CPU baseline, secondaries),
#                                               rocprofv3 kernel stats of the three F5 / BigVGAN configs, PMC passes, plumbing run of --gpus 2
#
# Replaces the 44 one-off tools/r3/call*.sh and the per-round rN_final.sh scripts.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-r5}
WHAT=${1:-bench}; shift || true
O=$ROOT/gpurun_out/$ROUND/$WHAT; mkdir -p $O
cd $ROOT
B="python $ROOT/bench.py"
Q="--no-secondary --no-cpu-baseline --no-pmc"
R="rocprofv3 --kernel-trace --stats --output-format csv"
# runb NAME [bench args]: stdout (the ONE compact line) -> $O/NAME.json, stderr -> $O/NAME.err, the full record -> $O/NAME_detail.json
runb() { local n=$1; shift; rm -f $ROOT/bench_detail.json; timeout 1500 $B "$@" > $O/$n.json 2> $O/$n.err; [ -f $ROOT/bench_detail.json ] && cp $ROOT/bench_detail.json $O/${n}_detail.json; summary $O/$n.json; }
summary() { python - "$1" <<'PY'
import json, os, sys
try:
    line = open(sys.argv[1]).read().strip().splitlines()[-1]
    d = json.loads(line)
    print(f"{sys.argv[1].split('/')[-1]}: compact line {len(line)} bytes")
    det = sys.argv[1][:-5] + "_detail.json"
    if os.path.exists(det):
        d = json.loads(open(det).read())
except Exception as e:
    print(sys.argv[1], "no JSON line:", e); sys.exit(0)
r = d.get("roofline") or {}
print(f"{sys.argv[1].split('/')[-1]}: {d['ms_per_step']:.2f} ms  {d['value']:.2f} {d['unit']}  n_gpus {d['n_gpus']}  {d['config'].get('arithmetic_kind')}  fold {d['config'].get('adaln_fold')}")
if r:
    print(f"   roofline {r['kernel']}: frac {r['frac']:.3f} ({r['achieved']:.1f} / {r['peak']:.1f} {r['unit']}), traffic {r.get('traffic')}")
    for k in r.get("kernels", [])[:8]:
        print(f"     {k['kernel'][:78]:78s} {k['ms_per_step']:8.2f} ms  x{k['launches_per_step']:7.0f}  {k['avg_launch_us']:7.1f} us")
for n, v in (d.get("secondary") or {}).items():
    print(f"   secondary {n}: {v.get('ms_per_step', v.get('ms_per_round_of_two', 0)):.2f} ms  {v['value']:.1f}")
c = d.get("cpu_baseline")
if c:
    print("   cpu_baseline", round(c["value"], 4), "cores", c["cores"], "evaluations", c.get("evaluations_run"))
PY
}
case $WHAT in
tests)
    timeout 2700 python -m pytest tests -m gpu -q -x -rA --timeout 900 "$@" > $O/tests_gpu_rA.log 2>&1; tail -12 $O/tests_gpu_rA.log ;;
bench)
    runb bench "$@" ;;
ab)
    A=$1; Bb=$2; N=${3:-2}
    for i in $(seq 1 $N); do
        runb a_$i $Q $A
        runb b_$i $Q $Bb
    done ;;
layers)
    cd /tmp; export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/t -- $B $Q --steps 2 --warmup 2 "$@" > /dev/null 2>&1
    python $ROOT/tools/x3p_by_shape.py $O/t | tee $O/layers.txt; rm -rf $O/t ;;
cpu-full)
    nproc > $O/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $O/host.txt
    timeout 2400 $B --cpu-baseline-only --cpu-baseline-full > $O/cpu_baseline_full.json 2> $O/err.log; tail -c 300 $O/cpu_baseline_full.json ;;
final)
    timeout 2700 python -m pytest tests -m gpu -q -x -rA --timeout 900 > $O/tests_gpu_rA.log 2>&1; tail -3 $O/tests_gpu_rA.log
    runb bench_default --steps 20 --warmup 5
    runb bench_bigvgan_f16_b8 --workload bigvgan
    runb bench_indextts --workload indextts --no-cpu-baseline
    MI355TTS_BENCH_BACKEND=gloo MI355TTS_BENCH_ONE_GPU=1 MI355TTS_BENCH_SMALL=1 runb bench_gpus2_selflaunch_plumbing --gpus 2 --steps 2 --warmup 1
    cd /tmp; export TMPDIR=/tmp
    timeout 600 $R -d $O/t_f5_f32_u1 -- $B $Q --steps 3 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_f32_under_rocprof.json
    timeout 600 $R -d $O/t_f5_bf16_u8 -- $B $Q --dtype bf16 --batch 8 --steps 2 --warmup 2 2>/dev/null | tail -1 > $O/bench_f5_bf16_u8_under_rocprof.json
    timeout 600 $R -d $O/t_bigvgan_f16_b8 -- $B $Q --workload bigvgan --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_bigvgan_f16_b8_under_rocprof.json
    for d in f5_f32_u1 f5_bf16_u8 bigvgan_f16_b8; do cp $O/t_$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
    rm -rf $O/t_*
    # PMC (separate passes, no trace domains beside --kernel-trace): MFMA busy / wave cycles, fabric traffic, L2 hit rate per kernel
    for W in "f5 f32 1 2" "bigvgan f16 8 1"; do
        set -- $W; P=$O/pmc_$1; mkdir -p $P
        if [ $1 = f5 ]; then C="python $ROOT/tools/pmc_f5_eval.py $2 $3 $4"; else C="python $ROOT/tools/pmc_bigvgan.py $2 $3 $4"; fi
        timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $P/p_sq -- $C > $P/sq.log 2>&1
        timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/p_fetch -- $C > $P/fetch.log 2>&1
        timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/p_write -- $C > $P/write.log 2>&1
        timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/p_tcc -- $C > $P/tcc.log 2>&1
        python $ROOT/tools/pmc_summary.py $P/p_sq $P/p_fetch $P/p_write $P/p_tcc > $O/$1_$2_pmc_by_kernel.json
        rm -rf $P
    done
    head -6 $O/f5_f32_u1_kernel_stats.csv | cut -c1-170 ;;
*) echo "unknown: $WHAT"; exit 2 ;;
esac
