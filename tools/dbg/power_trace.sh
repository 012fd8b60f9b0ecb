#!/bin/bash
# sample package power and shader clock while the bench runs: tools/dbg/power_trace.sh [bench args]
python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 40 --warmup 3 "$@" > /tmp/pt_bench.json &
BP=$!
sleep 20
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.5
done
wait $BP
python -c 'import json; d=json.loads(open("/tmp/pt_bench.json").read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"])'
