set -u
O=gpurun_out/r4_cpu; mkdir -p $O
nproc > $O/nproc.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $O/nproc.txt
timeout 2400 python bench.py --cpu-baseline-only --cpu-baseline-full > $O/cpu_baseline_full.json 2> $O/err.log
tail -c 400 $O/cpu_baseline_full.json; cat $O/nproc.txt
