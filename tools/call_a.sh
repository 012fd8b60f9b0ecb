set -u
O=gpurun_out/r4_a; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_limits.py -m gpu -q -rA --timeout 900 > $O/tests.log 2>&1; tail -45 $O/tests.log
