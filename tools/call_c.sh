set -u
ROOT=$PWD; O=$ROOT/gpurun_out/r4_c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f5.py -m gpu -q -x --timeout 900 -k "adaln or coexist or range_watch" 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 2 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t_fold -- $B > /dev/null 2>&1
python $ROOT/tools/x3p_by_shape.py $O/t_fold
rm -rf $O/t_fold
cd $ROOT
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc"
timeout 600 $B --steps 10 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold', d['ms_per_step'])"
timeout 600 $B --steps 10 --warmup 3 --no-adaln-fold | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nofold', d['ms_per_step'])"
