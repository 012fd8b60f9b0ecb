set -u
O=gpurun_out/r4_a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_f5.py -m gpu -q -x -rA --timeout 900 -k "multirank or bench or adaln or coexist or range_watch or bigvgan_type or mel_handoff or real_prompt or two_ranks or device_blob" > $O/tests.log 2>&1; tail -25 $O/tests.log
