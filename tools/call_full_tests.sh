set -u
O=gpurun_out/r4_tests; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -rA --timeout 900 > $O/tests_gpu.log 2>&1; tail -15 $O/tests_gpu.log
