set -u
O=gpurun_out/r4_base; mkdir -p $O
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["roofline"]["traffic"])
for k in d["roofline"]["kernels"][:8]: print("    ",k["kernel"],round(k["ms_per_step"],3),round(k["avg_launch_us"],1),round(k.get("tflops",0),1))
for n,v in d.get("secondary",{}).items(): print("   sec",n,round(v["ms_per_step"],1),round(v["value"],1))
PY
