cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests/test_gpu_bigvgan.py -x -q 2>&1 | tail -4 > gpurun_out/ab/t_bv.log
for i in 1 2; do for v in 1 0; do echo -n "AA_PIPE=$v "; MI355TTS_AA_PIPE=$v python bench.py --workload bigvgan --no-cpu-baseline --no-pmc 2>&1 | python -c '
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith("BENCH_DETAIL "):
        d=json.loads(l[13:]); print(round(d["ms_per_step"],3), [(k["kernel"][:22], k["launches_per_step"], round(k["avg_launch_us"],1)) for k in d["roofline"]["kernels"][:6]])
'; done; done > gpurun_out/ab/aa_pipe.txt 2>&1
cat gpurun_out/ab/t_bv.log gpurun_out/ab/aa_pipe.txt
