#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -q -rA -m gpu > gpurun_out/r3/pytest_full_x2.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3/pytest_full_x2.log; grep -E "^FAILED|^ERROR" gpurun_out/r3/pytest_full_x2.log | head -20
