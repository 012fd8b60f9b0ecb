#!/bin/bash
# round-2 measurement set A: determinism stress, GPU tests, default bench, BigVGAN with / without the 1-WG/CU aa_conv policy
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2
bash tools/ubench/aa_race.sh > gpurun_out/r2/aa_race.txt 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2/tests.log
python bench.py --workload bigvgan --no-cpu-baseline > gpurun_out/r2/bench_bigvgan_1wg.json 2> gpurun_out/r2/bench_bigvgan_1wg.err
MI355TTS_AACONV_LDS_MIN=0 python bench.py --workload bigvgan --no-cpu-baseline > gpurun_out/r2/bench_bigvgan_2wg.json 2> gpurun_out/r2/bench_bigvgan_2wg.err
python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
cat gpurun_out/r2/aa_race.txt; cat gpurun_out/r2/tests.log
python - <<'PY'
import json
for f in ("bench_bigvgan_1wg","bench_bigvgan_2wg","bench_default"):
    try:
        d=json.loads(open(f"gpurun_out/r2/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
        for k in d["roofline"]["kernels"][:6]: print("    ",k["kernel"],round(k["ms_per_step"],3),round(k["avg_launch_us"],1))
        if "secondary" in d:
            for n,v in d["secondary"].items(): print("   sec",n,v["ms_per_step"],v["value"])
        if "cpu_baseline" in d: print("   cpu",d["cpu_baseline"])
    except Exception as e: print(f,"ERR",e)
PY
