set -u
O=gpurun_out/r4_a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_f5.py -m gpu -q -x -rA --timeout 900 -k "adaln or coexist or range_watch or golden or end_to_end" > $O/tests.log 2>&1; tail -40 $O/tests.log
