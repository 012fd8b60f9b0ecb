#!/bin/bash
bash tools/dbg/x3d_gpu1.sh > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_gpu_compat.py tests/test_gpu_multirank.py -x -q > gpurun_out/r6/compat_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6/compat_pytest.txt
grep -v "^T[0-9]" gpurun_out/r6/x3d_check.txt; grep -c BAD gpurun_out/r6/x3d_check.txt; tail -5 gpurun_out/r6/x3d_pytest.txt; cat gpurun_out/r6/x3d_ab.txt | cut -c1-300; tail -30 gpurun_out/r6/compat_pytest.txt
